"""Two ranks (gloo rendezvous, both on the one GPU of the test box) running the REAL kernels through
vae_amd.dist: each rank owns a contiguous row shard, draws its noise in-kernel (Philox keyed by the global row),
and the packed scalar all-reduce turns the local sums into global-batch means.  The 2-rank result must equal the
1-rank result on the whole batch: per-sample outputs and gradients bit for bit (rows are independent and the
draw does not depend on the sharding), the means to rounding (different summation split)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BG, L, K, D = 203, 24, 12, 784        # ragged: 102 + 101 rows


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(BG, 2 * L + K, generator=g)
    x = (torch.rand(BG, 1, 28, 28, generator=g) < 0.13).float()
    wdec = torch.randn(L + K, D, generator=g) * 0.1
    return logits, x, wdec


def _run(rank, world, logits, x, wdec):
    """One rank's step on its shard (world == 1: the whole batch)."""
    from vae_amd import dist as VD, rng
    from vae_amd.elbo import ELBOMixin
    from vae_amd.reparameterizers import ConcatReparameterizer
    import torch.nn as nn

    class Model(ELBOMixin, nn.Module):
        def __init__(self):
            nn.Module.__init__(self)
            self.config = {'continuous_size': 2 * L, 'discrete_size': K, 'cuda': True, 'half': False,
                           'nll_type': 'bernoulli', 'kl_beta': 0.7, 'continuous_mut_info': 0, 'discrete_mut_info': 0}
            self.reparameterizer = ConcatReparameterizer(['isotropic_gaussian', 'discrete'], self.config)
    m = Model()
    m.train()
    rng.manual_seed(1234)
    if world > 1:
        row0, rows = VD.configure(m, BG)
    else:
        row0, rows = 0, BG
    sl = slice(row0, row0 + rows)
    lg = logits[sl].to(DEV).requires_grad_(True)
    z, params = m.reparameterizer(lg)
    recon = (z @ wdec.to(DEV)).view(rows, 1, 28, 28)
    out = m.loss_function(recon, x[sl].to(DEV), params)
    g = params['concat'][0]['gaussian']
    if world > 1:
        out, (mu_mean, lv_mean) = VD.globalize_loss(out, [g['mu_mean'], g['logvar_mean']])
    else:
        mu_mean, lv_mean = g['mu_mean'], g['logvar_mean']
    out['loss_mean'].backward()
    cpu = lambda t: t.detach().cpu()
    return {'row0': row0, 'rows': rows, 'z': cpu(z), 'loss': cpu(out['loss']), 'elbo': cpu(out['elbo']),
            'means': {k: cpu(out[k]) for k in ('loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean')},
            'mu_mean': cpu(mu_mean), 'lv_mean': cpu(lv_mean), 'grad': cpu(lg.grad)}


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    res = _run(rank, world, *_inputs())
    torch.save(res, os.path.join(outdir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_kernels_equal_one_rank(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    parts = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r), weights_only=True) for r in range(world)]
    one = _run(0, 1, *_inputs())
    assert [(p['row0'], p['rows']) for p in parts] == [(0, 102), (102, 101)]
    for p in parts:
        sl = slice(p['row0'], p['row0'] + p['rows'])
        assert torch.equal(p['z'], one['z'][sl])                      # same Philox draw whatever the sharding
        assert torch.equal(p['loss'], one['loss'][sl]) and torch.equal(p['elbo'], one['elbo'][sl])
        assert torch.equal(p['grad'], one['grad'][sl])                # backward needs no communication
        for k, v in p['means'].items():
            torch.testing.assert_close(v, one['means'][k], rtol=1e-6, atol=1e-5)
        torch.testing.assert_close(p['mu_mean'], one['mu_mean'], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(p['lv_mean'], one['lv_mean'], rtol=1e-5, atol=1e-7)


# ---- BatchNormFlow with the statistics of the whole sharded batch (vae_amd.dist.configure sets the hooks) ----
FB, FF = 203, 16


def _run_flow(rank, world):
    from vae_amd import dist as VD
    from vae_amd.flows import build_maf_flow
    g = torch.Generator().manual_seed(9)
    x = torch.randn(FB, FF, generator=g) * 1.5 + 0.3
    w = torch.randn(FB, FF, generator=g)
    torch.manual_seed(3)                       # the same flow weights on every rank
    flow = build_maf_flow(FF, 32, num_blocks=2, activation_str='tanh').to(DEV).train()
    for m in flow.modules():                   # non-trivial affine parameters
        if hasattr(m, 'log_gamma'):
            with torch.no_grad():
                m.log_gamma.copy_(torch.linspace(-0.3, 0.4, FF))
                m.beta.copy_(torch.linspace(0.5, -0.5, FF))
    row0, rows = VD.configure(flow, FB) if world > 1 else (0, FB)
    sl = slice(row0, row0 + rows)
    xi = x[sl].to(DEV).requires_grad_(True)
    u, logdet = flow(xi)
    ((u * w[sl].to(DEV)).sum() / FB + logdet.sum() / FB).backward()
    cpu = lambda t: t.detach().cpu()
    bn = [m for m in flow.modules() if hasattr(m, 'log_gamma')]
    return {'row0': row0, 'rows': rows, 'u': cpu(u), 'logdet': cpu(logdet), 'gx': cpu(xi.grad),
            'pgrads': [cpu(p.grad) for p in flow.parameters()],
            'running': [cpu(m.running_mean) for m in bn] + [cpu(m.running_var) for m in bn]}


def _flow_worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.save(_run_flow(rank, world), os.path.join(outdir, 'f%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_maf_flow_uses_whole_batch_statistics(tmp_path):
    """A 2-block MAF (MADE + BatchNormFlow + Reverse) on a batch sharded over two ranks equals the one-rank run:
    outputs, logdet and d x per row, running statistics, and parameter gradients summed over ranks."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_flow_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        if p.exitcode is None:
            p.kill()
        assert p.exitcode == 0
    parts = [torch.load(os.path.join(str(tmp_path), 'f%d.pt' % r), weights_only=True) for r in range(world)]
    one = _run_flow(0, 1)
    close = lambda a, b, tol=2e-5: torch.testing.assert_close(a, b, rtol=tol, atol=tol * max(1.0, float(b.abs().max())))
    for p in parts:
        sl = slice(p['row0'], p['row0'] + p['rows'])
        close(p['u'], one['u'][sl])
        close(p['logdet'], one['logdet'][sl])
        close(p['gx'], one['gx'][sl], 1e-4)
        for a, b in zip(p['running'], one['running']):
            close(a, b)
    for i, b in enumerate(one['pgrads']):
        close(parts[0]['pgrads'][i] + parts[1]['pgrads'][i], b, 1e-4)
