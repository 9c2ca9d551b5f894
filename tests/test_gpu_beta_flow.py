"""GPU parity: Beta (split + KL) and the Flow "KL" reduction vs golden vectors and the oracle."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def beta(L, train=True):
    from vae_amd.reparameterizers import Beta
    m = Beta({'continuous_size': 2 * L, 'cuda': True, 'half': False, 'continuous_mut_info': 0})
    m.train(train)
    return m


@pytest.mark.parametrize('name', ['beta_L4', 'beta_L32'])
def test_beta_golden(golden, name):
    g = golden('beta')[name]
    L = g['logits'].shape[1] // 2
    m = beta(L)
    logits = g['logits'].to(DEV).requires_grad_(True)
    z, params = m(logits)
    assert set(params) == {'z', 'logits', 'beta'}
    assert set(params['beta']) == {'conc1', 'conc2', 'conc1_mean', 'conc2_mean'}
    gb = g['params']['beta']
    for k in ('conc1', 'conc2', 'conc1_mean', 'conc2_mean'):
        close(params['beta'][k], gb[k], rtol=1e-6, atol=1e-7)
    assert z.shape == g['z_drawn'].shape and float(z.min()) >= 0 and float(z.max()) <= 1
    kl0 = m.kl(params)
    close(kl0, g['kl_none'], rtol=1e-5, atol=1e-4)
    pl = g['prior_logits'].to(DEV).requires_grad_(True)
    _, pparams = m(pl)
    klp = m.kl(params, pparams)
    close(klp, g['kl_prior'], rtol=1e-5, atol=1e-4)
    # log-likelihood of the reference's drawn sample under our params (the draw itself is not injectable)
    close(m.log_likelihood(g['z_drawn'].to(DEV), params), g['log_likelihood'], rtol=1e-4, atol=1e-4)
    (g0,) = torch.autograd.grad(kl0.sum() + params['beta']['conc1_mean'] + 2 * params['beta']['conc2_mean'], logits,
                                retain_graph=True)
    close(g0, g['grad_kl0_means_logits'], rtol=1e-4, atol=1e-5)
    gq, gp = torch.autograd.grad(klp.sum(), [logits, pl])
    close(gq, g['grad_klp_logits'], rtol=1e-4, atol=1e-5)
    close(gp, g['grad_klp_prior_logits'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('B,L', [(5, 3), (1000, 32), (64, 512), (300, 130)])
def test_beta_vs_oracle(B, L):
    torch.manual_seed(B + L)
    logits = torch.randn(B, 2 * L) * 3
    lc = logits.clone().requires_grad_(True)
    bp = O.beta_params(lc)
    k0 = O.beta_kl({'beta': bp})
    w = torch.randn(B)
    (g0,) = torch.autograd.grad((k0 * w).sum() + 5 * bp['conc1_mean'], lc)
    m = beta(L)
    lg = logits.to(DEV).requires_grad_(True)
    z, p = m(lg)
    k1 = m.kl(p)
    (g1,) = torch.autograd.grad((k1 * w.to(DEV)).sum() + 5 * p['beta']['conc1_mean'], lg, retain_graph=True)
    close(p['beta']['conc1'], bp['conc1'], rtol=1e-6, atol=1e-7)
    close(k1, k0, rtol=2e-5, atol=1e-4)
    # d KL / d logit = s(1-s) [(a - 1.1) psi'(a) + (2.2 - a1 - a0) psi'(a1 + a0)]: two O(1..1e3) terms of opposite
    # sign; fp32 (kernel and oracle alike) keeps ~1e-6 of the terms, not of their difference
    close(g1, g0, rtol=1e-4, atol=1e-4)
    # the sample's gradient reaches the logits through the kernel's backward (rsample is torch's)
    (gz,) = torch.autograd.grad(z.sum(), lg)
    assert torch.isfinite(gz).all() and float(gz.abs().sum()) > 0


@pytest.mark.parametrize('name', ['flowkl_2d', 'flowkl_3d'])
def test_flow_kl_golden(golden, name):
    from vae_amd import functional as F_
    g = golden('flowkl')[name]
    u = g['u'].to(DEV).requires_grad_(True)
    ld = g['logdet'].to(DEV).requires_grad_(True)
    kl = F_.flow_kl(u, ld)
    close(kl, g['kl'], rtol=1e-5, atol=1e-5)
    gu, gl = torch.autograd.grad((kl * g['w'].to(DEV)).sum(), [u, ld])
    close(gu, g['grad_u'], rtol=1e-5, atol=1e-6)
    close(gl, g['grad_logdet'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('shape', [(4096, 64), (1000, 6), (257, 1024), (64, 5, 32), (33, 3, 7)])
def test_flow_kl_vs_oracle(shape):
    from vae_amd import functional as F_
    torch.manual_seed(sum(shape))
    u = torch.randn(*shape)
    ld = torch.randn(shape[0], 1) if len(shape) == 2 else torch.randn(shape[0], shape[1])
    uc, lc = u.clone().requires_grad_(True), ld.clone().requires_grad_(True)
    k0 = O.flow_kl({'recon_x': uc, 'logdet': lc})
    w = torch.randn(shape[0])
    g0 = torch.autograd.grad((k0 * w).sum(), [uc, lc])
    ud, ldd = u.to(DEV).requires_grad_(True), ld.to(DEV).requires_grad_(True)
    k1 = F_.flow_kl(ud, ldd)
    g1 = torch.autograd.grad((k1 * w.to(DEV)).sum(), [ud, ldd])
    close(k1, k0, rtol=1e-5, atol=1e-4)
    close(g1[0], g0[0], rtol=1e-4, atol=1e-6)
    close(g1[1], g0[1], rtol=1e-5, atol=1e-6)


def test_flow_kl_full_batch_without_BxB():
    """B = 65536: the reference would materialise a 17 GB [B,B] tensor; O(B*F) here.  Property check:
    kl + logdet is the same scalar for every sample and equals mean(nlp)."""
    from vae_amd import functional as F_
    B, Fd = 65536, 64
    u = torch.randn(B, Fd, device=DEV)
    ld = torch.randn(B, 1, device=DEV)
    kl = F_.flow_kl(u, ld)
    c = kl + ld[:, 0]
    ref = (0.5 * u.double() ** 2).sum(-1).mean() + Fd * 0.9189385332046727
    assert float((c - c[0]).abs().max()) < 1e-4
    assert abs(float(c[0]) - float(ref)) < 1e-3
