"""CPU, world_size 2, gloo: the sharding + packed scalar all-reduce code path of vae_amd.dist, with the
oracle standing in for the kernels.  N-rank result == 1-rank result on the concatenated batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_loss(nll, kld, beta, inv_count):
    """What the fused kernel emits on a shard: per-sample loss/elbo + local sums * 1/B_global."""
    elbo = nll + kld
    loss = nll + beta * kld
    z = torch.zeros(())
    return {'loss': loss, 'elbo': elbo, 'loss_mean': loss.sum() * inv_count, 'elbo_mean': elbo.sum() * inv_count,
            'nll_mean': nll.sum() * inv_count, 'kld_mean': kld.sum() * inv_count, 'additional_loss_mean': z,
            'kl_beta_scalar': beta, 'mut_info_mean': z}


def _worker(rank, world, port, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vae_amd import dist as VD
    torch.manual_seed(0)
    Bg, L, F = 37, 8, 6            # ragged: 19 + 18 rows
    logits = torch.randn(Bg, 2 * L)
    noise = torch.randn(Bg, L)
    x = (torch.rand(Bg, 1, 8, 8) < 0.3).float()
    recon = torch.randn(Bg, 1, 8, 8)
    u, logdet = torch.randn(Bg, F), torch.randn(Bg, 1)
    row0, rows = VD.shard_rows(Bg)
    sl = slice(row0, row0 + rows)
    lg = logits[sl].clone().requires_grad_(True)
    z, params = O.gauss_forward(lg, noise[sl])
    kld = O.gauss_kl(params)
    nll = O.nll(x[sl], recon[sl], 'bernoulli')
    local = _local_loss(nll, kld, 0.5, 1.0 / Bg)
    mu_part = params['gaussian']['mu'].sum() / (Bg * L)
    lv_part = params['gaussian']['logvar'].sum() / (Bg * L)
    out, (mu_mean, lv_mean) = VD.globalize_loss(local, [mu_part, lv_part])
    out['loss_mean'].backward()
    # flow KL with the cross-batch mean all-reduced between the two phases
    nlp = (0.5 * u[sl] ** 2).sum(-1) + F * 0.9189385332046727
    mean = (nlp.sum() / Bg).reshape(1)
    VD.all_reduce_inplace(mean)
    flow_kl = mean - logdet[sl, 0]
    # results go through a file: tensors put on an mp.Queue travel as fds that die with this process
    torch.save((rank, row0, rows, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()},
                mu_mean.detach(), lv_mean.detach(), lg.grad.clone(), flow_kl), os.path.join(outdir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r), weights_only=True) for r in range(world)]

    torch.manual_seed(0)
    Bg, L, F = 37, 8, 6
    logits = torch.randn(Bg, 2 * L).requires_grad_(True)
    noise = torch.randn(Bg, L)
    x = (torch.rand(Bg, 1, 8, 8) < 0.3).float()
    recon = torch.randn(Bg, 1, 8, 8)
    u, logdet = torch.randn(Bg, F), torch.randn(Bg, 1)
    z, params = O.gauss_forward(logits, noise)
    ref = O.loss_function(O.nll(x, recon, 'bernoulli'), O.gauss_kl(params), 0.5)
    ref['loss_mean'].backward()
    ref_flow = O.flow_kl({'recon_x': u, 'logdet': logdet})

    assert [r[1:3] for r in res] == [(0, 19), (19, 18)]
    for rank, row0, rows, out, mu_mean, lv_mean, grad, flow_kl in res:
        for k in ('loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean'):
            torch.testing.assert_close(out[k], ref[k].detach(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(out['loss'], ref['loss'][row0:row0 + rows].detach(), rtol=1e-6, atol=1e-5)
        torch.testing.assert_close(mu_mean, params['gaussian']['mu_mean'].detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(lv_mean, params['gaussian']['logvar_mean'].detach(), rtol=1e-5, atol=1e-6)
        # per-rank gradients are the exact rows of the global-batch gradient (no communication in backward)
        torch.testing.assert_close(grad, logits.grad[row0:row0 + rows], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(flow_kl, ref_flow[row0:row0 + rows], rtol=1e-5, atol=1e-5)


def test_shard_rows_cover_batch():
    from vae_amd.dist import shard_rows
    for B in (0, 1, 7, 64, 65537):
        for W in (1, 2, 3, 8):
            spans = [shard_rows(B, r, W) for r in range(W)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == B
            for (a, n), (b, _) in zip(spans, spans[1:]):
                assert a + n == b


def test_single_process_is_identity():
    from vae_amd.dist import packed_all_reduce
    a, b = torch.tensor(1.5), torch.arange(3.)
    x, y = packed_all_reduce([a, b])
    assert x is a and y is b


def _grad_worker(rank, world, port, outdir, buckets=1):
    """A tiny encoder -> (oracle) Gaussian latent + Bernoulli loss -> decoder model on this rank's rows: the flat
    gradient buffer summed over ranks must be the single-process gradient of the global batch."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vae_amd import dist as VD
    torch.manual_seed(1)
    Bg, L, Dp = 23, 4, 16
    enc, dec = torch.nn.Linear(Dp, 2 * L), torch.nn.Linear(L, Dp)
    x = (torch.rand(Bg, Dp) < 0.4).float()
    noise = torch.randn(Bg, L)
    row0, rows = VD.shard_rows(Bg)
    sl = slice(row0, row0 + rows)
    flat = VD.FlatGradients(list(enc.parameters()) + list(dec.parameters()), buckets=buckets)
    assert len(flat._pieces) == 1 if buckets == 1 else 1 < len(flat._pieces) <= buckets
    for _ in range(2):              # twice: the second pass must not see the first pass's sums
        flat.zero()
        z, params = O.gauss_forward(enc(x[sl]), noise[sl])
        local = _local_loss(O.nll(x[sl], dec(z), 'bernoulli'), O.gauss_kl(params), 1.0, 1.0 / Bg)
        out, _ = VD.globalize_loss(local)
        out['loss_mean'].backward()
        flat.all_reduce()
    assert all(p.grad.untyped_storage().data_ptr() == flat.flat.untyped_storage().data_ptr() for p in flat.params)
    torch.save((flat.flat.clone(), out['loss_mean'].detach()), os.path.join(outdir, 'g%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('buckets', [1, 2, 4])
def test_flat_gradients_two_ranks_equal_one_rank(tmp_path, buckets):
    """buckets > 1: the pieces are all-reduced asynchronously from post-accumulate-grad hooks while the backward pass
    is still running (the decoder's piece first); same sums."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, str(tmp_path), buckets)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(str(tmp_path), 'g%d.pt' % r), weights_only=True) for r in range(world)]
    torch.manual_seed(1)
    Bg, L, Dp = 23, 4, 16
    enc, dec = torch.nn.Linear(Dp, 2 * L), torch.nn.Linear(L, Dp)
    x = (torch.rand(Bg, Dp) < 0.4).float()
    noise = torch.randn(Bg, L)
    z, params = O.gauss_forward(enc(x), noise)
    ref = O.loss_function(O.nll(x, dec(z), 'bernoulli'), O.gauss_kl(params), 1.0)
    ref['loss_mean'].backward()
    want = torch.cat([p.grad.reshape(-1) for p in list(enc.parameters()) + list(dec.parameters())])
    for flat, loss_mean in res:
        torch.testing.assert_close(flat, want, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(loss_mean, ref['loss_mean'].detach(), rtol=1e-5, atol=1e-5)


def test_flat_gradients_refuses_replaced_grads():
    import pytest
    from vae_amd import dist as VD
    lin = torch.nn.Linear(3, 2)
    flat = VD.FlatGradients(lin.parameters())
    assert flat.nbytes == 4 * (6 + 2)
    lin(torch.randn(5, 3)).sum().backward()
    assert float(flat.flat.abs().sum()) > 0
    flat.zero()
    assert float(flat.flat.abs().sum()) == 0
    lin.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        flat.zero()
