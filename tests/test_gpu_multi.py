"""GPU parity: the multi-sample kernels (S Gaussian draws from one logits read; S-sample NLL + ELBO reading
x once) against the reference's own draw loops (tests/golden/multi.pt) and the oracle."""
import pytest
import torch
import torch.nn as nn

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def cfg(**kw):
    c = {'continuous_size': 8, 'discrete_size': 5, 'cuda': True, 'half': False, 'epochs': 10,
         'continuous_mut_info': 0, 'discrete_mut_info': 0, 'kl_beta': 1.0, 'use_prior_kl': False,
         'reparam_type': 'isotropic_gaussian', 'nll_type': 'bernoulli', 'max_time_steps': 3}
    c.update(kw)
    return c


def make_model(g, mixin):
    from vae_amd.reparameterizers import IsotropicGaussian

    class Model(mixin, nn.Module):
        def __init__(self):
            super().__init__()
            self.config = cfg()
            self.reparameterizer = IsotropicGaussian(self.config)
            self.dec = nn.Linear(4, 12)
            self.enc = nn.Linear(12, 8)

        def decode(self, z):
            return self.dec(z)

        def encode(self, x):
            return self.enc(x)

    m = Model().to(DEV)
    with torch.no_grad():
        m.dec.weight.copy_(g['dec_weight'])
        m.dec.bias.copy_(g['dec_bias'])
        if 'enc_weight' in g:
            m.enc.weight.copy_(g['enc_weight'])
            m.enc.bias.copy_(g['enc_bias'])
    return m


def test_loss_function_k3_golden(golden):
    """ELBOMixin.loss_function(K=3) against AbstractVAE.loss_function(K=3) run by the reference: the first
    draw through forward(), the two further draws through the multi-sample kernel, the three decoder outputs
    through one NLL+ELBO launch.  Noise is injected by patching the two draw entry points."""
    from vae_amd import elbo as E
    g = golden('multi')['multi_lossK3']
    m = make_model(g, E.ELBOMixin).train()
    logits = g['logits'].to(DEV).requires_grad_(True)
    noises = [n.to(DEV) for n in g['noises']]
    rep = m.reparameterizer
    z0, params = rep(logits, noise=noises[0])
    fm = rep.forward_multi
    rep.forward_multi = lambda lg, S, force=False, noise=None: fm(lg, S, force=force, noise=noises[1:1 + S])
    res = m.loss_function(m.decode(z0), g['x'].to(DEV), params, K=g['K'])
    for k, v in g['result'].items():
        if torch.is_tensor(v):
            close(res[k], v, rtol=1e-5, atol=1e-5)
        else:
            assert res[k] == v
    gl, gw, gb = torch.autograd.grad(res['loss_mean'], [logits, m.dec.weight, m.dec.bias])
    close(gl, g['grad_logits'], rtol=1e-4, atol=1e-6)
    close(gw, g['grad_dec_weight'], rtol=1e-4, atol=1e-6)
    close(gb, g['grad_dec_bias'], rtol=1e-4, atol=1e-6)


def test_msg_reparameterize_golden(golden):
    from vae_amd import elbo as E
    g = golden('multi')['multi_msg']
    m = make_model({'dec_weight': torch.zeros(12, 4), 'dec_bias': torch.zeros(12)}, E.MSGMixin).train()
    logits = g['logits'].to(DEV).requires_grad_(True)
    z_list, p_list = m.reparameterizer.forward_multi(logits, 3, noise=[n.to(DEV) for n in g['noises']])
    assert len(z_list) == 3 and len(p_list) == 3
    for z, zr in zip(z_list, g['z_list']):
        assert torch.equal(z.detach().cpu(), zr)                     # bit-exact draws
        assert z.is_contiguous()
    assert set(p_list[-1]) == set(g['params_last']) and set(p_list[-1]['gaussian']) == set(g['params_last']['gaussian'])
    assert torch.equal(p_list[-1]['gaussian']['logvar'].detach().cpu(), g['params_last']['gaussian']['logvar'])
    kl = m.kld(p_list)
    close(kl, g['kl'], rtol=1e-5, atol=1e-5)
    w = g['w'].to(DEV)
    (gl,) = torch.autograd.grad(sum((z * wi).sum() for z, wi in zip(z_list, w)) + kl.sum(), logits)
    close(gl, g['grad_logits'], rtol=1e-5, atol=1e-5)
    # the mixin's own reparameterize(): same shapes, Philox draws, distinct per draw, reproducible
    import vae_amd
    vae_amd.rng.manual_seed(3)
    za, _ = m.reparameterize(logits)
    vae_amd.rng.manual_seed(3)
    zb = [m.reparameterizer(logits)[0] for _ in range(3)]             # three single calls: same streams
    assert all(torch.equal(a, b) for a, b in zip(za, zb))
    assert not torch.equal(za[0], za[1])


def _draw_major(noises, B, P):
    """The reference draws sample i's pass p as noises[i * P + p] ([B, L]: one row per replica of the sample);
    the batched path wants [S = P * B, B, L], draw-major."""
    L = noises[0].shape[-1]
    out = torch.empty(P * B, B, L)
    for i in range(B):
        for p in range(P):
            out[p * B:(p + 1) * B, i] = noises[i * P + p]
    return out


@pytest.mark.parametrize('path', ['batched', 'loop'])
@pytest.mark.parametrize('name', ['multi_likelihood_draws', 'multi_likelihood_mean'])
def test_likelihood_golden(golden, name, path):
    """AbstractVAE.likelihood as the reference ran it (tests/golden/multi.pt) against both forms here: the batched
    one (all draws of the minibatch from one logits read, one decoder call, log-sum-exp folded into the scoring
    launch's closing kernel) and the reference's loop, which an overridden reparameterize() selects."""
    from vae_amd import elbo as E
    g = golden('multi')[name]
    B, K = g['x'].shape[0], g['K']
    mixin = E.ELBOMixin
    if path == 'loop':
        class mixin(E.ELBOMixin):
            def reparameterize(self, logits, labels=None, force=False):
                return self.reparameterizer(logits, force=force)
    m = make_model(g, mixin).eval()
    m.reparameterizer.train(g['reparam_training'])
    loader = [(g['x'], torch.zeros(B).long())]
    if not g['reparam_training']:
        val = m.likelihood(loader, K=K)
    elif path == 'loop':
        noises = iter([n.to(DEV) for n in g['noises']])
        fwd = m.reparameterizer.forward
        m.reparameterizer.forward = lambda lg, force=False, noise=None: fwd(lg, force=force, noise=next(noises))
        val = m.likelihood(loader, K=K)
    else:
        val = m.likelihood(loader, K=K, noise=[_draw_major(g['noises'], B, K // B).to(DEV)])
    close(val, g['value'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('rows', [1 << 16, 300, 7])     # one decoder call / sample groups / a lone sample in pieces
def test_likelihood_batched_equals_loop_on_the_same_draws(rows):
    """B = 12 samples, K = 150 -> 12 passes, S = 144 draws per sample (two scoring launches of <= 128 draws each,
    combined), against the reference's loop fed the same noise; also a ragged last minibatch."""
    from vae_amd import elbo as E

    class Loop(E.ELBOMixin):
        def reparameterize(self, logits, labels=None, force=False):
            return self.reparameterizer(logits, force=force)

    torch.manual_seed(5)
    g = {'dec_weight': torch.randn(12, 4) * 0.5, 'dec_bias': torch.randn(12) * 0.1,
         'enc_weight': torch.randn(8, 12) * 0.5, 'enc_bias': torch.randn(8) * 0.1}
    K = 150
    loader, per_batch, flat = [], [], []
    for B in (12, 5):
        P = K // B
        x = (torch.rand(B, 12) < 0.4).float()
        noises = [torch.randn(B, 4) for _ in range(B * P)]
        loader.append((x, torch.zeros(B).long()))
        per_batch.append(_draw_major(noises, B, P).to(DEV))
        flat += noises
    fast = make_model(g, E.ELBOMixin).eval()
    fast.reparameterizer.train(True)
    fast.config['vae_amd_likelihood_rows'] = rows
    slow = make_model(g, Loop).eval()
    slow.reparameterizer.train(True)
    it = iter([n.to(DEV) for n in flat])
    fwd = slow.reparameterizer.forward
    slow.reparameterizer.forward = lambda lg, force=False, noise=None: fwd(lg, force=force, noise=next(it))
    close(fast.likelihood(loader, K=K, noise=per_batch), slow.likelihood(loader, K=K), rtol=1e-5, atol=1e-5)
    # Philox draws: reproducible, and close to the injected-noise estimate (same estimator, other draws)
    import vae_amd
    vae_amd.rng.manual_seed(9)
    a = fast.likelihood(loader, K=K)
    vae_amd.rng.manual_seed(9)
    b = fast.likelihood(loader, K=K)
    assert torch.equal(a, b) and bool(torch.isfinite(a))


@pytest.mark.parametrize('B,L,S', [(1, 1, 1), (7, 3, 2), (64, 128, 4), (1000, 36, 3), (257, 512, 5)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_gauss_multi_matches_single_calls(B, L, S, dtype):
    """S draws in one launch == S single-draw launches on the same noise: z bit-equal, KL/means equal,
    d(logits) equal up to the summation order over draws."""
    from vae_amd import functional as F_
    torch.manual_seed(B + S)
    logits = torch.randn(B, 2 * L, device=DEV).to(dtype)
    noise = torch.randn(S, B, L, device=DEV).to(dtype)
    w = torch.randn(S, B, L, device=DEV).to(dtype)
    wk = torch.randn(B, device=DEV).to(dtype)

    def run(multi, up=False):
        x = (logits.float() if up else logits.clone()).requires_grad_(True)
        if multi:
            z, lv, kl, means = F_.gauss_reparam_multi(x, S, noise=noise)
            zs = list(z)
        else:
            outs = [F_.gauss_reparam_kl(x, noise=noise[s].float() if up else noise[s]) for s in range(S)]
            zs, lv, kl, means = [o[0] for o in outs], outs[0][1], outs[0][2], outs[0][3]
        obj = sum((zi.float() * wi.float()).sum() for zi, wi in zip(zs, w)) + (kl.float() * wk.float()).sum() \
            + means[0] + 2 * means[1] + (lv.float() * 0.125).sum()
        (g,) = torch.autograd.grad(obj, x)
        return zs, lv, kl, means, g

    z1, lv1, kl1, m1, g1 = run(True)
    z0, lv0, kl0, m0, g0 = run(False)
    for a, b in zip(z1, z0):
        assert torch.equal(a, b)
    assert torch.equal(lv1, lv0)
    close(kl1, kl0, rtol=1e-6 if dtype == torch.float32 else 1e-2, atol=1e-6)   # same terms, other summation order
    close(m1, m0, rtol=1e-6, atol=1e-6)
    if dtype == torch.float32:
        close(g1, g0, rtol=1e-5, atol=1e-5)
    else:
        # the single-call path adds S bf16-rounded gradients in bf16; judge the one-launch gradient (fp32
        # accumulation, one rounding) against the same computation on the upcast inputs instead
        g32 = run(False, up=True)[4]
        if dtype == torch.float16:       # d KL / d s = s - 1/s beyond fp16's range: inf, as torch's own cast gives
            big = g32.abs() > 6e4
            assert bool(torch.isinf(g1[big]).all())
            g1, g32 = g1.float().masked_fill(big, 0), g32.masked_fill(big, 0)
        close(g1, g32, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize('nll_type', ['bernoulli', 'gaussian', 'disc_logistic'])
@pytest.mark.parametrize('B,D,S', [(1, 1, 1), (5, 784, 3), (300, 75, 8), (17, 12288, 2), (64, 784, 11)])
def test_nll_multi_vs_oracle(nll_type, B, D, S):
    from vae_amd import functional as F_
    torch.manual_seed(B + D + S)
    R = D if nll_type == 'bernoulli' else 2 * D
    x = (torch.rand(B, D) < 0.13).float() if nll_type == 'bernoulli' else torch.floor(torch.rand(B, D) * 256) / 255.0
    recons = [torch.randn(B, R) * 1.5 for _ in range(S)]
    kld, add, mi = torch.rand(B) * 10, torch.rand(B), torch.rand(B)
    w = torch.randn(B)

    r0 = [r.clone().requires_grad_(True) for r in recons]
    k0 = kld.clone().requires_grad_(True)
    n0 = O.nll_multi(x, r0, nll_type)
    d0 = O.loss_function(n0, k0, 0.7, mut_info=mi, extra={'a': add})
    obj0 = (d0['loss'] * w).sum() + 0.5 * (d0['elbo'] * w).sum() + d0['loss_mean'] + 2 * d0['nll_mean'] + d0['elbo_mean']
    g0 = torch.autograd.grad(obj0, r0 + [k0])

    r1 = [r.to(DEV).requires_grad_(True) for r in recons]
    k1 = kld.to(DEV).requires_grad_(True)
    n1, loss, elbo, means = F_.nll_multi_elbo(x.to(DEV), r1, nll_type, kld=k1, add=add.to(DEV), mi=mi.to(DEV), kl_beta=0.7)
    close(n1, n0, rtol=1e-5, atol=1e-6 * D)
    close(loss, d0['loss'], rtol=1e-5, atol=1e-6 * D)
    close(elbo, d0['elbo'], rtol=1e-5, atol=1e-6 * D)
    close(means[0], d0['loss_mean'], rtol=1e-5, atol=1e-6 * D)
    close(means[2], d0['nll_mean'], rtol=1e-5, atol=1e-6 * D)
    wd = w.to(DEV)
    obj1 = (loss * wd).sum() + 0.5 * (elbo * wd).sum() + means[0] + 2 * means[2] + means[1]
    g1 = torch.autograd.grad(obj1, r1 + [k1])
    if nll_type == 'bernoulli':
        for a, b in zip(g1, g0):
            close(a, b, rtol=1e-5, atol=2e-6)
    else:
        # ill-conditioned elements (see test_gpu_elbo.close_cond): aggregate accuracy
        for a, b in zip(g1[:-1], g0[:-1]):
            err = (a.detach().cpu().double() - b.double()).norm() / max(float(b.double().norm()), 1e-30)
            assert float(err) < 2e-4, float(err)
        close(g1[-1], g0[-1], rtol=1e-5, atol=1e-6)


def test_nll_multi_rejects_cpu_and_bad_shapes():
    from vae_amd import functional as F_
    with pytest.raises(RuntimeError):
        F_.nll_multi_elbo(torch.zeros(2, 4), [torch.zeros(2, 4)], 'bernoulli')
    with pytest.raises(ValueError):
        F_.nll_multi_elbo(torch.zeros(2, 4, device=DEV), [torch.zeros(2, 5, device=DEV)], 'bernoulli')


@pytest.mark.parametrize('dtype,rtol', [(torch.float16, 4e-3), (torch.bfloat16, 2e-2)])
def test_nll_multi_more_than_max_samples_16bit(dtype, rtol):
    """S > VAE_MAX_SAMPLES in a 16-bit element type: group launches + functional.elbo_combine, whose backward round 1
    dispatched as bf16 whatever the dtype."""
    from vae_amd import functional as F_
    torch.manual_seed(3)
    B, D, S = 64, 96, 11
    x = (torch.rand(B, D) < 0.3).to(dtype)
    recons = [(torch.randn(B, D) * 1.5).to(dtype) for _ in range(S)]
    kld = (torch.rand(B) * 4).to(dtype)
    r0 = [r.float().requires_grad_(True) for r in recons]
    k0 = kld.float().requires_grad_(True)
    d0 = O.loss_function(O.nll_multi(x.float(), r0, 'bernoulli'), k0, 0.7)
    g0 = torch.autograd.grad(d0['loss_mean'] + 0.5 * d0['elbo_mean'], r0 + [k0])
    r1 = [r.to(DEV).requires_grad_(True) for r in recons]
    k1 = kld.to(DEV).requires_grad_(True)
    n1, loss, elbo, means = F_.nll_multi_elbo(x.to(DEV), r1, 'bernoulli', kld=k1, kl_beta=0.7)
    close(loss, d0['loss'], rtol=rtol, atol=rtol)
    close(means[0], d0['loss_mean'], rtol=rtol, atol=rtol)
    g1 = torch.autograd.grad(means[0] + 0.5 * means[1], r1 + [k1])
    for a, b in zip(g1, g0):
        assert a.dtype == dtype
        close(a, b, rtol=rtol, atol=rtol * float(b.abs().max()))


@pytest.mark.parametrize('dtype,rtol', [(torch.float32, 1e-5), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)])
@pytest.mark.parametrize('T,B', [(1, 9), (5, 300), (128, 33)])
def test_steps_logsumexp_epilogue(dtype, rtol, T, B):
    """with_lse: the closing launch's log-sum-exp over the steps == torch.logsumexp over the per-step elbo column it
    also returns; +inf / -inf entries behave as torch's (max shift taken out only when finite)."""
    from vae_amd import functional as F_
    torch.manual_seed(T + B)
    D = 40
    x = (torch.rand(B, D, device=DEV) < 0.4).to(dtype)
    recons = [(torch.randn(B, D, device=DEV) * 2).to(dtype) for _ in range(T)]
    kld = (torch.rand(T * B, device=DEV) * 3).to(dtype)
    out = F_.nll_steps_elbo([x] * T, recons, 'bernoulli', kld=kld, with_lse=True)
    elbo, lse = out[2], out[6]
    assert lse.shape == (B,) and lse.dtype == dtype and not lse.requires_grad
    want = torch.logsumexp(elbo.float().view(T, B), 0)
    close(lse, want, rtol=rtol, atol=rtol)
    assert len(F_.nll_steps_elbo([x] * T, recons, 'bernoulli', kld=kld)) == 6
