"""Batch sharding of the hot path over the GPUs of one node (one process per GPU, RCCL over xGMI).

Every function on the path is per-sample (row) independent, so ranks own contiguous row
ranges and no data-path collective exists.  What does cross ranks:

  * the batch scalars -- the six ELBO means (abstract_vae.py:381-387), the Gaussian / Beta
    `*_mean` diagnostics -- as ONE packed all-reduce(sum) of <= 10 floats per step.  Each rank's
    kernels already scale their local sums by 1/B_global (`inv_count`, `mean_scale`), so the sum
    over ranks IS the global-batch mean; the backward needs no communication.
  * the Flow "KL" cross-batch mean (flow.py:78 / SURVEY Q6): one extra float, all-reduced between
    the kernel's two phases (`reduce_fn`).
  * gradients of the stock encoder / decoder parameters: `FlatGradients` below -- every parameter's
    .grad is a view of ONE flat buffer, all-reduced (sum) in place as one message after backward
    (the VAE networks are a few MB, so one flat bucket beats many: the reduction is
    latency-bound on xGMI) -- or the host project's DDP.  With 1/B_global scaling the per-rank
    gradients SUM to the exact global-batch gradient; under DDP's mean reduction multiply the loss
    by `world_size`.

Noise: in-kernel Philox is keyed by the GLOBAL row index (`row_offset`), so an N-rank run draws
exactly what the 1-rank run draws for the same rows.

Works on any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def _many(group=None):
    """True when collectives have to run: an initialised process group of more than one rank -- or of exactly one with
    VAE_AMD_DIST_FORCE=1, which sends every collective of this module through the backend anyway (the only way to run
    the RCCL side on a one-GPU box: stream waits, async work objects and hooks are the same with one rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get('VAE_AMD_DIST_FORCE', '0') == '1'


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
    if not _many(group):
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
    if _many(group):
        _sum_over_ranks_(t, group)
    return t


class FlatGradients(object):
    """One flat gradient buffer for a set of parameters: `p.grad` of every parameter is a view into it (autograd
    accumulates into an existing .grad in place), so the cross-rank sum is ONE all-reduce of the buffer where it
    lies -- no bucketing, no copies, the same code under RCCL ('nccl') and under gloo (CPU tests, several ranks
    rehearsing on one GPU: staged through the host by _sum_over_ranks_).

        flat = FlatGradients(model.parameters())
        ... flat.zero(); loss.backward(); flat.all_reduce(); optimizer.step()      # never zero_grad(set_to_none=True)

    `buckets` > 1 (larger networks): the buffer is cut into that many contiguous pieces of about equal size and a
    piece is all-reduced, asynchronously on the collective's own stream, as soon as the backward pass has accumulated
    the gradient of its last parameter (post-accumulate-grad hooks) -- the pieces holding the decoder's parameters
    cross xGMI while the encoder's backward still runs; all_reduce() then only launches what is left and waits.
    One backward() per zero() / all_reduce() pair (no gradient accumulation over several backward passes).  Under a
    gloo group with device tensors the pieces are reduced at all_reduce() (host-staged, synchronous)."""

    def __init__(self, params, buckets=1):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no parameters that require a gradient"
        p0 = self.params[0]
        sizes = [p.numel() for p in self.params]
        total = sum(sizes)
        self.flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        o = 0
        starts = []
        for p, n in zip(self.params, sizes):
            assert p.dtype == p0.dtype and p.device == p0.device, "one dtype and device per FlatGradients"
            p.grad = self.flat[o:o + n].view_as(p)
            starts.append(o)
            o += n
        # contiguous pieces of about total / buckets elements, cut at parameter boundaries
        buckets = max(1, min(int(buckets), len(self.params)))
        self._pieces = []          # (first element, end element, number of parameters)
        first, count, target = 0, 0, total / buckets
        for k, (st, n) in enumerate(zip(starts, sizes)):
            count += 1
            last = k == len(sizes) - 1
            if last or (st + n - first >= target and len(self._pieces) < buckets - 1):
                self._pieces.append((first, st + n, count))
                first, count = st + n, 0
        self._owner = {}
        self._pending = [c for _, _, c in self._pieces]
        self._work = [None] * len(self._pieces)
        self._launched = [False] * len(self._pieces)
        self._group = None
        if len(self._pieces) > 1:
            k = 0
            for idx, (p, st) in enumerate(zip(self.params, starts)):
                while st >= self._pieces[k][1]:
                    k += 1
                self._owner[idx] = k
                p.register_post_accumulate_grad_hook(lambda _p, piece=k: self._accumulated(piece))

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def _distributed(self):
        return _many(self._group)

    def _async_ok(self):
        # a gloo group reduces device tensors through the host, synchronously (see _sum_over_ranks_): at all_reduce()
        return not (self.flat.is_cuda and dist.get_backend(self._group) == 'gloo')

    def _launch(self, k):
        a, b, _ = self._pieces[k]
        self._launched[k] = True
        if self._async_ok():
            self._work[k] = dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self._group, async_op=True)
        else:
            _sum_over_ranks_(self.flat[a:b], self._group)

    def _accumulated(self, k):
        self._pending[k] -= 1
        if self._pending[k] == 0 and self._distributed() and self._async_ok():
            self._launch(k)

    def zero(self):
        for p in self.params:       # somebody replaced a view (zero_grad(set_to_none=True), a fresh .grad): put it back
            if p.grad is None or p.grad.untyped_storage().data_ptr() != self.flat.untyped_storage().data_ptr():
                raise RuntimeError("FlatGradients: a parameter's .grad no longer aliases the flat buffer "
                                   "(use flat.zero(), not zero_grad(set_to_none=True))")
        assert not any(w is not None for w in self._work), "FlatGradients: zero() before all_reduce() finished the step"
        self.flat.zero_()
        self._pending = [c for _, _, c in self._pieces]
        self._launched = [False] * len(self._pieces)

    def all_reduce(self, group=None):
        """Sum over ranks, in place; identity when not distributed.  With buckets: launches the pieces the backward
        pass has not already sent and waits for all of them (the current stream waits; the host does not block under
        RCCL)."""
        if len(self._pieces) == 1:
            return all_reduce_inplace(self.flat, group)
        assert group is self._group or group is None, "bucketed FlatGradients: pass the group to set_group() first"
        if self._distributed():
            for k in range(len(self._pieces)):
                if not self._launched[k]:
                    self._launch(k)
            for k, w in enumerate(self._work):
                if w is not None:
                    w.wait()
                    self._work[k] = None
        return self.flat

    def set_group(self, group):
        self._group = group


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
