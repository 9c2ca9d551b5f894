"""HIP graph capture of the hot path through the drop-in classes: every launch is asynchronous on torch's
current stream, nothing synchronises or allocates outside torch's (capturable) allocator, so a whole
forward + loss + backward records into one graph.  Draws made while capturing stay in-kernel Philox draws: their
call offset lives in a device word that a one-thread node of the graph advances (vae_rng_next, vae_amd/rng.py), so
every replay gets fresh noise and the backward kernels regenerate exactly the noise of their replay.

Checked: replayed outputs equal the oracle evaluated on the noise recovered from the replay's own z;
two replays draw different noise; the replayed gradients match the oracle's."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def test_graph_capture_gaussian_elbo_step():
    from vae_amd.reparameterizers import IsotropicGaussian
    from vae_amd.elbo import loss_function
    torch.manual_seed(0)
    B, L = 256, 32
    logits = torch.randn(B, 2 * L)
    logits[:, L:] = logits[:, L:].abs() + 0.5          # scale away from 0 so the noise can be recovered from z
    x = (torch.rand(B, 1, 28, 28) < 0.13).float()
    wdec = torch.randn(L, 784) * 0.1
    m = IsotropicGaussian({'continuous_size': 2 * L, 'cuda': True, 'half': False})
    m.train()
    lg = logits.to(DEV).requires_grad_(True)
    xd, wd = x.to(DEV), wdec.to(DEV)

    def step():
        z, params = m(lg)
        recon = (z @ wd).view(B, 1, 28, 28)             # a stock "decoder"
        out = loss_function(recon, xd, m.kl(params), 'bernoulli')
        (g,) = torch.autograd.grad(out['loss_mean'], lg)
        return z, out['loss_mean'], out['kld_mean'], g

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):                              # warm-up on the side stream (allocator, workspace)
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        z, loss_mean, kld_mean, g = step()
    seen = []
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        zc = z.detach().cpu()
        e = (zc - logits[:, :L]) / (0.5 * (logits[:, L:] + 1e-6))     # the draw this replay used
        seen.append(e)
        assert abs(float(e.mean())) < 0.05 and abs(float(e.std()) - 1.0) < 0.05
        lc = logits.clone().requires_grad_(True)
        z0, p0 = O.gauss_forward(lc, e)
        close(z0, zc, rtol=1e-5, atol=1e-5)
        n0 = O.nll(x, (z0 @ wdec).view(B, 1, 28, 28), 'bernoulli')
        r0 = O.loss_function(n0, O.gauss_kl(p0))
        close(loss_mean, r0['loss_mean'], rtol=1e-5, atol=1e-3)
        close(kld_mean, r0['kld_mean'], rtol=1e-5, atol=1e-4)
        (g0,) = torch.autograd.grad(r0['loss_mean'], lc)
        close(g, g0, rtol=1e-4, atol=1e-6)
    assert not torch.equal(seen[0], seen[1])            # fresh noise on every replay
    # reproducible: restarting the counters replays the same sequence of draws
    import vae_amd
    vae_amd.rng.manual_seed(11)
    graph.replay()
    za = z.detach().clone()
    graph.replay()
    zb = z.detach().clone()
    vae_amd.rng.manual_seed(11)
    graph.replay()
    assert torch.equal(z, za) and not torch.equal(za, zb)
    # no draw was materialised: the graph holds no torch.randn node's [B, L] buffer (the draw is in-kernel); its
    # offset word comes from the per-device pool that the warm-up steps allocated OUTSIDE the capture, and the pool
    # hands its words out again after release_captured()
    pool = vae_amd.rng._pools[torch.cuda.current_device()]
    assert pool[1] >= 1 and not vae_amd.rng._cells
    used = pool[1]
    del graph
    vae_amd.rng.release_captured()
    assert pool[1] == 0 and used <= vae_amd.rng._POOL_WORDS


def test_graph_capture_gumbel_and_vrnn_pair():
    from vae_amd.reparameterizers import GumbelSoftmax, IsotropicGaussian
    from vae_amd import vrnn as V
    torch.manual_seed(1)
    B, K, L = 128, 16, 8
    gm = GumbelSoftmax({'discrete_size': K, 'cuda': True, 'half': False})
    gm.train()
    rep = IsotropicGaussian({'continuous_size': 2 * L, 'cuda': True, 'half': False})
    rep.train()
    cfg = {'reparam_type': 'isotropic_gaussian', 'use_prior_kl': True, 'continuous_size': 2 * L}
    lgum = torch.randn(B, K, device=DEV)
    enc, pri = torch.randn(B, 2 * L, device=DEV), torch.randn(B, 2 * L, device=DEV)

    def step():
        y, p = gm(lgum)
        z, params = V.reparameterize(rep, {'encoder_logits': enc, 'prior_logits': pri.clone(), 'x_features': None}, cfg)
        return y, p['discrete']['z_hard'], z['posterior'], z['prior'], V.kld(rep, params, cfg)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    it0 = gm.iteration
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y, zh, zq, zp, kl = step()
    outs = []
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        outs.append((y.clone(), zq.clone(), zp.clone()))
        close(y.sum(-1), torch.ones(B), rtol=1e-5, atol=1e-5)
        assert bool(((zh == 1).sum(-1) >= 1).all()) and bool(torch.isfinite(kl).all())
    assert gm.iteration == it0 + 1                      # host state advances at capture time only
    for a, b in zip(outs[0], outs[1]):
        assert not torch.equal(a, b)


def test_capture_helper_concat_step_with_static_input():
    """vae_amd.graph.capture: a Concat(Gaussian, Gumbel) step (one-launch row kernel, in-kernel noise) recorded once,
    replayed on NEW data copied into the static input: the replay follows the data (KL of the new logits against the
    oracle), draws fresh noise every time, and its gradient matches the oracle on the noise recovered from z."""
    import vae_amd
    from vae_amd.reparameterizers import ConcatReparameterizer
    torch.manual_seed(2)
    B, L, K = 512, 32, 64
    m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'],
                              {'continuous_size': 2 * L, 'discrete_size': K, 'cuda': True, 'half': False, 'epochs': 10})
    m.train()
    static = torch.zeros(B, 2 * L + K, device=DEV, requires_grad=True)
    w = torch.randn(B, L + K, device=DEV)

    def step():
        z, params = m(static)
        kl = m.kl(params)
        (g,) = torch.autograd.grad((z * w).sum() + kl.sum(), static)
        return z, kl, g

    cap = vae_amd.graph.capture(step)
    seen = []
    for trial in range(2):
        logits = torch.randn(B, 2 * L + K)
        logits[:, L:2 * L] = logits[:, L:2 * L].abs() + 0.5
        with torch.no_grad():
            static.copy_(logits.to(DEV))
        for rep in range(2):
            z, kl, g = cap()
            torch.cuda.synchronize()
            zc = z.detach().cpu()
            e = (zc[:, :L] - logits[:, :L]) / (0.5 * (logits[:, L:2 * L] + 1e-6))
            seen.append(e)
            assert abs(float(e.mean())) < 0.05 and abs(float(e.std()) - 1.0) < 0.05
            p0 = O.gauss_kl({'gaussian': {'mu': logits[:, :L], 'logvar': logits[:, L:2 * L] + 1e-6}})
            close(kl, p0, rtol=1e-5, atol=1e-4)       # the Gumbel KL against no prior is identically 0
            # the soft sample is a probability row
            close(z[:, L:].sum(-1), torch.ones(B), rtol=1e-5, atol=1e-5)
            assert bool(torch.isfinite(g).all())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[2], seen[3])
    cap.release()
    vae_amd.rng.release_captured()
