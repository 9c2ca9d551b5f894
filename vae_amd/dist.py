"""Batch sharding of the hot path over the GPUs of one node (one process per GPU, RCCL over xGMI).

Every function on the path is per-sample (row) independent, so ranks own contiguous row
ranges and no data-path collective exists.  What does cross ranks:

  * the batch scalars -- the six ELBO means (abstract_vae.py:381-387), the Gaussian / Beta
    `*_mean` diagnostics -- as ONE packed all-reduce(sum) of <= 10 floats per step.  Each rank's
    kernels already scale their local sums by 1/B_global (`inv_count`, `mean_scale`), so the sum
    over ranks IS the global-batch mean; the backward needs no communication.
  * the Flow "KL" cross-batch mean (flow.py:78 / SURVEY Q6): one extra float, all-reduced between
    the kernel's two phases (`reduce_fn`).
  * gradients of the stock encoder / decoder parameters: the host project's DDP (not here).  With
    1/B_global scaling the per-rank gradients SUM to the exact global-batch gradient; under DDP's
    mean reduction multiply the loss by `world_size`.

Noise: in-kernel Philox is keyed by the GLOBAL row index (`row_offset`), so an N-rank run draws
exactly what the 1-rank run draws for the same rows.

Works on any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_rows(global_batch, rank=None, world_size=None):
    """(row0, rows) of this rank's contiguous shard; the first `global_batch % world` ranks get one more."""
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    base, extra = divmod(global_batch, world_size)
    rows = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    return row0, rows


def _sum_over_ranks_(t, group=None):
    """In-place all-reduce(sum).  RCCL reduces device tensors where they are; a gloo group (CPU tests, several
    ranks rehearsing on one GPU) gets the few floats through the host."""
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class _AllReduceSum(torch.autograd.Function):
    """sum over ranks; the gradient passes through unchanged (every rank seeds the same cotangent)."""

    @staticmethod
    def forward(ctx, t, group):
        return _sum_over_ranks_(t.clone(), group)

    @staticmethod
    def backward(ctx, g):
        return g, None


def packed_all_reduce(tensors, group=None):
    """All-reduce(sum) several small tensors as ONE message; returns tensors of the original shapes.
    A no-op (identity) when torch.distributed is not initialised or the world is one rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(tensors)
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    flat = _AllReduceSum.apply(flat, group)
    out, o = [], 0
    for t in tensors:
        n = t.numel()
        out.append(flat[o:o + n].reshape(t.shape).to(t.dtype))
        o += n
    return out


def all_reduce_inplace(t, group=None):
    """reduce_fn for functional.flow_kl: in-place sum of a small float tensor over ranks."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _sum_over_ranks_(t, group)
    return t


def configure(module, global_batch, rank=None, world_size=None, group=None):
    """Point every reparameterizer under `module` (and an ELBOMixin model itself) at this rank's shard:
    global row offset for Philox, 1/B_global scaling for the means, the flow-KL reduction hook."""
    row0, rows = shard_rows(global_batch, rank, world_size)
    for m in module.modules():
        if hasattr(m, 'row_offset'):
            m.row_offset = row0
        if hasattr(m, 'mean_scale') and hasattr(m, 'output_size'):
            m.mean_scale = 1.0 / (global_batch * m.output_size)
        if hasattr(m, 'reduce_fn'):
            m.reduce_fn = lambda t, g=group: all_reduce_inplace(t, g)
        if hasattr(m, 'inv_count'):
            m.inv_count = 1.0 / global_batch
    return row0, rows


MEAN_KEYS = ('loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'additional_loss_mean', 'mut_info_mean')


def globalize_loss(loss_dict, extra_scalars=(), group=None):
    """Turn a loss dict whose `*_mean` entries are local sums / B_global (what the kernels emit when
    `inv_count = 1/B_global`) into global-batch means, with one packed all-reduce that also carries
    `extra_scalars` (e.g. mu_mean, logvar_mean).  Returns (loss_dict, extra_scalars)."""
    keys = [k for k in MEAN_KEYS if k in loss_dict]
    red = packed_all_reduce([loss_dict[k] for k in keys] + list(extra_scalars), group)
    out = dict(loss_dict)
    for k, v in zip(keys, red):
        out[k] = v
    return out, red[len(keys):]
