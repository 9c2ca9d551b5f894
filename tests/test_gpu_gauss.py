"""GPU parity: IsotropicGaussian HIP path vs the oracle and the reference's golden vectors.

Tolerances (north_star): z bit-exact under injected noise (same two roundings); KL / means
rtol 1e-5 fp32; bf16 rtol 1e-2.
"""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def mod(L, train=True):
    from vae_amd.reparameterizers import IsotropicGaussian
    m = IsotropicGaussian({'continuous_size': 2 * L, 'cuda': True, 'half': False})
    m.train(train)
    return m


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


@pytest.mark.parametrize('name', ['gauss_L4_train', 'gauss_L4_eval', 'gauss_L32_train', 'gauss_L32_eval'])
def test_golden(golden, name):
    g = golden('gauss')[name]
    L = g['logits'].shape[1] // 2
    m = mod(L, g['training'])
    logits = g['logits'].to(DEV).requires_grad_(True)
    z, params = m(logits, noise=g['noise'].to(DEV))
    assert set(params) == {'z', 'logits', 'gaussian'}
    assert set(params['gaussian']) == {'z', 'mu', 'logvar', 'mu_mean', 'logvar_mean'}
    assert torch.equal(z.detach().cpu(), g['z'])                    # bit-exact
    gp = g['params']['gaussian']
    assert torch.equal(params['gaussian']['logvar'].detach().cpu(), gp['logvar'])
    assert torch.equal(params['gaussian']['mu'].detach().cpu(), gp['mu'])
    close(params['gaussian']['mu_mean'], gp['mu_mean'])
    close(params['gaussian']['logvar_mean'], gp['logvar_mean'])
    kl = m.kl(params)
    close(kl, g['kl'], rtol=1e-5, atol=1e-4)
    # prior form (VRNN / class-conditioned)
    pl = g['prior_logits'].to(DEV).requires_grad_(True)
    _, pparams = m(pl, noise=g['noise'].to(DEV))
    klp = m.kl(params, pparams)
    close(klp, g['kl_prior'], rtol=1e-5, atol=1e-4)
    (gl,) = torch.autograd.grad((z * g['w'].to(DEV)).sum() + kl.sum(), logits, retain_graph=True)
    close(gl, g['grad_logits'], rtol=1e-5, atol=1e-5)
    gq, gpp = torch.autograd.grad(klp.sum(), [logits, pl])
    close(gq, g['grad_klp_logits'], rtol=1e-5, atol=1e-5)
    close(gpp, g['grad_klp_prior_logits'], rtol=1e-5, atol=1e-5)


def test_golden_3d(golden):
    g = golden('gauss')['gauss_3d']
    m = mod(8)
    z, params = m(g['logits'].to(DEV), noise=g['noise'].to(DEV))
    assert torch.equal(z.cpu(), g['z'])
    kl = m.kl(params)
    assert kl.shape == g['kl'].shape
    close(kl, g['kl'], atol=1e-4)


def test_golden_4d(golden):
    """[B, 2C, H, W] logits split on channels, read where they are (vae_gauss_reparam_kl_fwd/bwd_nchw): the reference's
    own run."""
    g = golden('gauss')['gauss_4d']
    m = mod(3)
    logits = g['logits'].to(DEV).requires_grad_(True)
    z, params = m(logits, noise=g['noise'].to(DEV))
    assert torch.equal(z.detach().cpu(), g['z'])
    gp = params['gaussian']
    assert torch.equal(gp['logvar'].detach().cpu(), g['params']['gaussian']['logvar'])
    assert torch.equal(gp['mu'].detach().cpu(), g['params']['gaussian']['mu'])
    close(gp['mu_mean'], g['params']['gaussian']['mu_mean'], atol=1e-6)
    close(gp['logvar_mean'], g['params']['gaussian']['logvar_mean'], atol=1e-6)
    kl = m.kl(params)
    assert kl.shape == g['kl'].shape
    close(kl, g['kl'], atol=1e-5)
    (gl,) = torch.autograd.grad((z * g['w'].to(DEV)).sum() + (kl * g['wk'].to(DEV)).sum(), logits)
    close(gl, g['grad_logits'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('shape', [(5, 4, 3, 7), (64, 16, 8, 8), (33, 2, 1, 128), (9, 6, 4, 36)])
@pytest.mark.parametrize('train', [True, False])
def test_4d_vs_oracle(shape, train):
    """Ragged and vector-aligned W, training and eval (z = the mu view), values and d(logits) of the whole tensor."""
    torch.manual_seed(sum(shape))
    B, C2, H, W = shape
    C = C2 // 2
    logits, noise = torch.randn(*shape), torch.randn(B, C, H, W)
    w, wk = torch.randn(B, C, H, W), torch.randn(B, C, H)
    lc = logits.clone().requires_grad_(True)
    z0, p0 = O.gauss_forward(lc, noise, training=train)
    k0 = O.gauss_kl(p0)
    (g0,) = torch.autograd.grad((z0 * w).sum() + (k0 * wk).sum() + p0['gaussian']['mu_mean'], lc)
    m = mod(C)
    m.train(train)
    ld = logits.to(DEV).requires_grad_(True)
    z1, p1 = m(ld, noise=noise.to(DEV))
    k1 = m.kl(p1)
    (g1,) = torch.autograd.grad((z1 * w.to(DEV)).sum() + (k1 * wk.to(DEV)).sum() + p1['gaussian']['mu_mean'], ld)
    assert torch.equal(z1.detach().cpu(), z0.detach())
    close(k1, k0, rtol=1e-5, atol=1e-4)
    close(g1, g0, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('B,L', [(1, 1), (3, 5), (7, 32), (64, 512), (1000, 100), (4096, 1024), (513, 2052)])
@pytest.mark.parametrize('train', [True, False])
def test_vs_oracle(B, L, train):
    torch.manual_seed(B * 31 + L)
    logits = torch.randn(B, 2 * L)
    noise = torch.randn(B, L)
    w = torch.randn(B, L)
    lc = logits.clone().requires_grad_(True)
    z0, p0 = O.gauss_forward(lc, noise, training=train)
    k0 = O.gauss_kl(p0)
    obj0 = (z0 * w).sum() + (k0 * 0.5).sum() + 3 * p0['gaussian']['mu_mean'] - 2 * p0['gaussian']['logvar_mean'] \
        + (p0['gaussian']['logvar'] * w).sum()
    (g0,) = torch.autograd.grad(obj0, lc)

    m = mod(L, train)
    lg = logits.to(DEV).requires_grad_(True)
    z1, p1 = m(lg, noise=noise.to(DEV))
    k1 = m.kl(p1)
    wd = w.to(DEV)
    obj1 = (z1 * wd).sum() + (k1 * 0.5).sum() + 3 * p1['gaussian']['mu_mean'] - 2 * p1['gaussian']['logvar_mean'] \
        + (p1['gaussian']['logvar'] * wd).sum()
    (g1,) = torch.autograd.grad(obj1, lg)
    assert torch.equal(z1.detach().cpu(), z0.detach())
    assert torch.equal(p1['gaussian']['logvar'].detach().cpu(), p0['gaussian']['logvar'].detach())
    close(k1, k0, rtol=1e-5, atol=1e-5 * L)
    close(p1['gaussian']['mu_mean'], p0['gaussian']['mu_mean'], atol=1e-6)
    close(p1['gaussian']['logvar_mean'], p0['gaussian']['logvar_mean'], atol=1e-6)
    close(g1, g0, rtol=1e-5, atol=1e-5)
    if not train:
        assert z1.data_ptr() == lg.data_ptr()  # eval returns the mu view itself (reference :96)


def test_strided_logits_view():
    """Concat hands the Gaussian child a narrow() view: row stride > 2L."""
    torch.manual_seed(1)
    B, L, K = 33, 8, 5
    full = torch.randn(B, 2 * L + K)
    noise = torch.randn(B, L)
    z0, p0 = O.gauss_forward(full[:, :2 * L], noise)
    m = mod(L)
    fd = full.to(DEV)
    z1, p1 = m(fd[:, :2 * L], noise=noise.to(DEV))
    assert torch.equal(z1.cpu(), z0)
    close(m.kl(p1), O.gauss_kl(p0), atol=1e-4)


def test_standalone_kl_vs_oracle():
    torch.manual_seed(2)
    B, L = 257, 48
    mq, sq, mp = torch.randn(B, L), torch.randn(B, L), torch.randn(B, L)
    sp = torch.rand(B, L) + 0.1
    from vae_amd import functional as F_
    args = [t.to(DEV).requires_grad_(True) for t in (mq, sq, mp, sp)]
    kl = F_.gauss_kl(*args)
    cargs = [t.clone().requires_grad_(True) for t in (mq, sq, mp, sp)]
    kl0 = O.gauss_kl({'gaussian': {'mu': cargs[0], 'logvar': cargs[1]}}, {'gaussian': {'mu': cargs[2], 'logvar': cargs[3]}})
    close(kl, kl0, rtol=1e-5, atol=1e-3)
    w = torch.randn(B)
    g1 = torch.autograd.grad((kl * w.to(DEV)).sum(), args)
    g0 = torch.autograd.grad((kl0 * w).sum(), cargs)
    for a, b in zip(g1, g0):
        close(a, b, rtol=1e-4, atol=1e-4)


def test_philox_noise_statistics_and_shard_invariance():
    """In-kernel Philox: N(0,1) moments, reproducible under manual_seed, independent of sharding."""
    from vae_amd import functional as F_
    B, L = 4096, 256
    logits = torch.cat([torch.zeros(B, L), torch.full((B, L), 2.0)], -1).to(DEV)  # z = e * (0.5*(2+1e-6))
    z, _, _, _ = F_.gauss_reparam_kl(logits, None, training=True, seed=1234, offset=1)
    e = z.float()
    assert abs(float(e.mean())) < 5e-3 and abs(float(e.std()) - 1.0) < 5e-3
    assert abs(float((e ** 4).mean()) - 3.0) < 0.1
    z2, _, _, _ = F_.gauss_reparam_kl(logits, None, training=True, seed=1234, offset=1)
    assert torch.equal(z, z2)
    zs, _, _, _ = F_.gauss_reparam_kl(logits[B // 2:], None, training=True, seed=1234, offset=1, row0=B // 2)
    assert torch.equal(zs, z[B // 2:])
    z3, _, _, _ = F_.gauss_reparam_kl(logits, None, training=True, seed=1234, offset=2)
    assert not torch.equal(z, z3)
    # backward regenerates the same draw: d z / d s = 0.5 e
    lg = logits.clone().requires_grad_(True)
    zz, _, _, _ = F_.gauss_reparam_kl(lg, None, training=True, seed=1234, offset=1)
    (g,) = torch.autograd.grad(zz.sum(), lg)
    close(g[:, L:], 0.5 * e, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('half', [torch.bfloat16, torch.float16])
def test_bf16(half):
    torch.manual_seed(3)
    B, L = 128, 64
    logits = torch.randn(B, 2 * L).to(half)
    noise = torch.randn(B, L).to(half)
    z0, p0 = O.gauss_forward(logits.float(), noise.float())
    m = mod(L)
    z1, p1 = m(logits.to(DEV), noise=noise.to(DEV))
    assert z1.dtype == half
    close(z1, z0, rtol=1e-2, atol=1e-2)
    close(m.kl(p1), O.gauss_kl(p0), rtol=1e-2, atol=1e-2 * L)


def test_empty_batch():
    m = mod(4)
    z, p = m(torch.zeros(0, 8, device=DEV), noise=torch.zeros(0, 4, device=DEV))
    assert z.shape == (0, 4) and m.kl(p).shape == (0,)


def test_logvar_is_produced_on_first_read():
    """params['gaussian']['logvar'] (isotropic_gaussian.py:124-125) is not written by the forward launch; the first
    read runs the reference's own op and yields the same bits, with gradient to the logits; kl() never needs it."""
    from vae_amd.lazy import LazyParams
    torch.manual_seed(4)
    B, L = 64, 24
    logits, noise = torch.randn(B, 2 * L), torch.randn(B, L)
    m = mod(L)
    lg = logits.to(DEV).requires_grad_(True)
    z, params = m(lg, noise=noise.to(DEV))
    g = params['gaussian']
    assert isinstance(g, LazyParams) and g.is_pending('logvar')
    assert list(g) == ['z', 'mu', 'logvar', 'mu_mean', 'logvar_mean']       # the reference's key order
    kl = m.kl(params)
    assert g.is_pending('logvar')                                            # the fused KL did not need it
    z0, p0 = O.gauss_forward(logits, noise)
    close(kl, O.gauss_kl(p0), atol=1e-4)
    lv = g['logvar']
    assert not g.is_pending('logvar') and g['logvar'] is lv
    assert torch.equal(lv.detach().cpu(), p0['gaussian']['logvar'])
    (gl,) = torch.autograd.grad(lv.sum(), lg)
    assert torch.equal(gl[:, :L].cpu(), torch.zeros(B, L)) and torch.equal(gl[:, L:].cpu(), torch.ones(B, L))
    # a plain copy of the params (what ConcatReparameterizer's {**child_params} does) reads it too
    z2, params2 = m(lg, noise=noise.to(DEV))
    flat = {**params2['gaussian']}
    assert type(flat) is dict and torch.equal(flat['logvar'].detach().cpu(), p0['gaussian']['logvar'])
    # the prior form of kl() reads both logvars
    klp = m.kl(params2, params)
    close(klp, torch.zeros(B), atol=1e-5)
