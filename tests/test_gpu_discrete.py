"""GPU parity: GumbelSoftmax / categorical KL / Bernoulli HIP paths vs the reference's golden vectors
and the oracle.  Hard samples (multi-hot masks and their argmax indices) are compared bit-exactly;
soft samples, KL and gradients at rtol 1e-5 (fp32)."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def gumbel(K, train=True, tau=1.0):
    from vae_amd.reparameterizers import GumbelSoftmax
    m = GumbelSoftmax({'discrete_size': K, 'cuda': True, 'half': False, 'discrete_mut_info': 0})
    m.train(train)
    m.tau = tau
    return m


GUMBEL = ['gumbel_K%d_%s_tau%g' % (K, m, t) for K in (5, 32) for m in ('train', 'eval') for t in (1.0, 0.5)]


@pytest.mark.parametrize('name', GUMBEL)
def test_gumbel_golden(golden, name):
    g = golden('gumbel')[name]
    K = g['logits'].shape[1]
    m = gumbel(K, g['training'], g['tau'])
    logits = g['logits'].to(DEV).requires_grad_(True)
    z, params = m(logits, noise=g['U'].to(DEV))
    assert set(params) == {'z', 'discrete'}
    assert set(params['discrete']) == {'z_hard', 'logits', 'log_q_z', 'tau_scalar'}
    d, gd = params['discrete'], g['params']['discrete']
    zh = d['z_hard'].detach().cpu()
    assert torch.equal(zh == 1, gd['z_hard'] == 1)                       # the hard mask, bit-exact
    assert torch.equal(zh.argmax(-1), g['hard_index'])
    assert torch.equal((zh == 1).sum(-1), g['hard_count'])
    close(params['z'], g['params']['z'])
    close(z, g['z'])
    if not g['training']:
        assert torch.equal(z.detach().cpu() == 1, g['z'] == 1)
    close(d['log_q_z'], gd['log_q_z'], atol=1e-5)
    assert float(m.kl(params).abs().max()) == 0.0
    pl = g['prior_logits'].to(DEV).requires_grad_(True)
    prior = {'discrete': {'log_q_z': torch.log_softmax(pl, -1)}}
    klp = m.kl(params, prior)
    close(klp, g['kl_prior'], atol=1e-5)
    gl, gp = torch.autograd.grad((z * g['w'].to(DEV)).sum() + klp.sum(), [logits, pl])
    close(gl, g['grad_logits'], rtol=1e-4, atol=1e-5)
    close(gp, g['grad_prior_logits'], rtol=1e-4, atol=1e-5)


def test_gumbel_ties_multi_hot(golden):
    g = golden('gumbel')['gumbel_ties']
    m = gumbel(8, False, 1.0)
    z, params = m(g['logits'].to(DEV), noise=g['U'].to(DEV))
    zh = params['discrete']['z_hard'].cpu()
    assert torch.equal(zh == 1, g['params']['discrete']['z_hard'] == 1)
    assert torch.equal((zh == 1).sum(-1), g['hard_count']) and int(g['hard_count'].max()) >= 2
    assert torch.equal(z.cpu() == 1, g['z'] == 1)


def test_gumbel_anneal(golden):
    g = golden('gumbel')['gumbel_anneal']
    m = gumbel(4, True)
    m.iteration = g['start_iteration']
    taus = []
    x = torch.zeros(2, 4, device=DEV)
    for _ in range(25):
        m(x)
        taus.append(float(m.tau))
    assert m.iteration == g['end_iteration'] and taus == g['taus']
    assert m.get_reparameterizer_scalars() == {'tau_scalar': m.tau}


@pytest.mark.parametrize('B,K', [(1, 1), (7, 3), (4096, 32), (2048, 256), (1024, 1024), (513, 10), (64, 1000),
                                 (37, 2048), (16, 4096), (9, 1023)])
@pytest.mark.parametrize('tau', [1.0, 0.7])
def test_gumbel_vs_oracle(B, K, tau):
    torch.manual_seed(B + K)
    logits = torch.randn(B, K) * 2
    U = torch.rand(B, K)
    w = torch.randn(B, K)
    lc = logits.clone().requires_grad_(True)
    z0, p0 = O.gumbel_forward(lc, U, tau=tau, training=True)
    d0 = p0['discrete']
    obj0 = (z0 * w).sum() + (d0['z_hard'] * w).sum() * 0.5 + (d0['log_q_z'] * w).sum()
    (g0,) = torch.autograd.grad(obj0, lc)
    m = gumbel(K, True, tau)
    lg = logits.to(DEV).requires_grad_(True)
    z1, p1 = m(lg, noise=U.to(DEV))
    d1 = p1['discrete']
    wd = w.to(DEV)
    obj1 = (z1 * wd).sum() + (d1['z_hard'] * wd).sum() * 0.5 + (d1['log_q_z'] * wd).sum()
    (g1,) = torch.autograd.grad(obj1, lg)
    h1, h0 = d1['z_hard'].detach().cpu() == 1, d0['z_hard'].detach() == 1
    assert torch.equal(h1, h0)                                           # every hard sample, bit-exact
    assert torch.equal(h1.float().argmax(-1), h0.float().argmax(-1))
    close(z1, z0, rtol=1e-5, atol=1e-7)
    close(d1['log_q_z'], d0['log_q_z'], rtol=1e-5, atol=1e-5)
    close(g1, g0, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('B,K', [(5, 4100), (3, 5000), (2, 4099), (2, 1025), (9, 20000)])
def test_gumbel_rows_beyond_the_register_resident_limit(B, K):
    """K > VAE_MAX_SOFTMAX_ROW (or > 1024 unaligned): the streamed three-pass kernels; the reference has no limit
    (gumbel.py:110-124).  Same checks as the resident path, hard samples bit-exact."""
    test_gumbel_vs_oracle(B, K, 0.7)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_gumbel_long_rows_16bit(dtype):
    B, K = 4, 8200
    torch.manual_seed(1)
    logits = (torch.randn(B, K) * 2).to(dtype)
    U = torch.rand(B, K).to(dtype)
    m = gumbel(K, True, 1.0)
    z1, p1 = m(logits.to(DEV), noise=U.to(DEV))
    # the reference rounds every op of the sample path in the tensor dtype (gumbel.py:269-277): the matching oracle is
    # the restatement run IN that dtype (as tests/test_gpu_bf16.py does for the register-resident kernels)
    z0, p0 = O.gumbel_forward(logits, U.float(), tau=1.0, training=True)
    assert z1.dtype == dtype
    z1f, z0f = z1.float().cpu(), z0.float()
    assert float((z1f - z0f).abs().max()) <= 2e-2 * float(z0f.abs().max())
    close(p1['discrete']['log_q_z'].float(), p0['discrete']['log_q_z'].float(), rtol=2e-2, atol=6e-2)
    hot = p1['discrete']['z_hard'].float().cpu() == 1
    assert int(hot.sum(-1).min()) >= 1
    ys = z1f
    assert bool((ys[hot] == ys.max(-1, keepdim=True)[0].expand_as(ys)[hot]).all())   # hot entries are row maxima of y


def test_gumbel_philox():
    from vae_amd import functional as F_
    B, K = 8192, 16
    logits = torch.zeros(B, K, device=DEV)
    y, zh, lq = F_.gumbel_softmax(logits, None, tau=1.0, seed=7, offset=3)
    idx = zh.argmax(-1)
    counts = torch.bincount(idx, minlength=K).float() / B
    assert float((counts - 1.0 / K).abs().max()) < 0.02                  # uniform logits -> uniform argmax
    y2, zh2, _ = F_.gumbel_softmax(logits[B // 2:], None, tau=1.0, seed=7, offset=3, row0=B // 2)
    assert torch.equal(zh2, zh[B // 2:]) and torch.equal(y2, y[B // 2:])  # shard-invariant draw


def test_categorical_kl_vs_oracle():
    from vae_amd import functional as F_
    torch.manual_seed(0)
    for B, K in [(33, 5), (1000, 32), (64, 1024), (5, 3000), (6, 6000), (3, 4099), (2, 1025)]:
        a = torch.log_softmax(torch.randn(B, K) * 2, -1)
        b = torch.log_softmax(torch.randn(B, K), -1)
        ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        k0 = O.categorical_kl({'discrete': {'log_q_z': ac}}, {'discrete': {'log_q_z': bc}})
        w = torch.randn(B)
        g0 = torch.autograd.grad((k0 * w).sum(), [ac, bc])
        ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        k1 = F_.categorical_kl(ad, bd)
        g1 = torch.autograd.grad((k1 * w.to(DEV)).sum(), [ad, bd])
        close(k1, k0, rtol=1e-5, atol=1e-5)
        close(g1[0], g0[0], rtol=1e-4, atol=1e-5)
        close(g1[1], g0[1], rtol=1e-4, atol=1e-5)


def bern(K, train=True, tau=1.0):
    from vae_amd.reparameterizers import Bernoulli
    m = Bernoulli({'discrete_size': K, 'cuda': True, 'half': False, 'epochs': 10})
    m.train(train)
    m.tau = tau
    return m


BERN = ['bern_K%d_%s_tau%g' % (K, m, t) for K in (5, 32) for m in ('train', 'eval') for t in (1.0, 0.3)]


@pytest.mark.parametrize('name', BERN)
def test_bernoulli_golden(golden, name):
    g = golden('bern')[name]
    K = g['logits'].shape[1]
    m = bern(K, g['training'], g['tau'])
    logits = g['logits'].to(DEV).requires_grad_(True)
    z, params = m(logits, noise=g['U'].to(DEV))
    assert set(params) == {'z', 'logits', 'discrete'} and set(params['discrete']) == {'z_hard', 'logits', 'tau_scalar'}
    close(params['z'], g['params']['z'], rtol=1e-5, atol=1e-6)
    # the hard sample: identical wherever the relaxed value is not within rounding of the 0.5 threshold
    ref_relaxed = g['params']['z']
    decided = (ref_relaxed - 0.5).abs() > 1e-5
    zh = params['discrete']['z_hard'].detach().cpu().round()
    assert torch.equal(zh[decided], g['params']['discrete']['z_hard'].round()[decided])
    kl0 = m.kl(params)
    close(kl0, g['kl_none'], atol=1e-5)
    pl = g['prior_logits'].to(DEV).requires_grad_(True)
    klp = m.kl(params, {'discrete': {'logits': pl}})
    close(klp, g['kl_prior'], atol=1e-5)
    (gl,) = torch.autograd.grad((z * g['w'].to(DEV)).sum() + kl0.sum(), logits, retain_graph=True)
    close(gl, g['grad_logits'], rtol=1e-4, atol=1e-5)
    gq, gp = torch.autograd.grad(klp.sum(), [logits, pl])
    close(gq, g['grad_klp_logits'], rtol=1e-4, atol=1e-5)
    close(gp, g['grad_klp_prior_logits'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('B,K', [(3, 1), (1000, 32), (257, 513), (64, 4096)])
def test_bernoulli_vs_oracle(B, K):
    torch.manual_seed(B * K)
    logits = torch.randn(B, K) * 3
    logits[0, 0] = 40.0       # saturates sigmoid: the clamp passes no gradient
    U = torch.rand(B, K)
    w = torch.randn(B, K)
    lc = logits.clone().requires_grad_(True)
    z0, p0 = O.bernoulli_forward(lc, U, tau=0.6)
    k0 = O.bernoulli_kl(p0)
    (g0,) = torch.autograd.grad((z0 * w).sum() + (p0['discrete']['z_hard'] * w).sum() + k0.sum(), lc)
    m = bern(K, True, 0.6)
    lg = logits.to(DEV).requires_grad_(True)
    z1, p1 = m(lg, noise=U.to(DEV))
    k1 = m.kl(p1)
    (g1,) = torch.autograd.grad((z1 * w.to(DEV)).sum() + (p1['discrete']['z_hard'] * w.to(DEV)).sum() + k1.sum(), lg)
    close(z1, z0, rtol=1e-5, atol=1e-6)
    decided = (z0.detach() - 0.5).abs() > 1e-5
    assert torch.equal(p1['discrete']['z_hard'].detach().cpu().round()[decided],
                       p0['discrete']['z_hard'].detach().round()[decided])
    close(k1, k0, rtol=1e-5, atol=1e-5)
    close(g1, g0, rtol=1e-4, atol=1e-5)


def test_training_forward_writes_the_soft_sample_only():
    """GumbelSoftmax / Bernoulli in training: z_hard (and log_q_z) keep their place in the params dict but are
    produced on first read, with the bits the eager (eval-mode) launch writes and with gradients to the logits."""
    from vae_amd.lazy import LazyParams
    from vae_amd.reparameterizers import GumbelSoftmax, Bernoulli
    from vae_amd import functional as F_
    torch.manual_seed(8)
    B, K = 96, 40
    logits, U = torch.randn(B, K), torch.rand(B, K)
    lg = logits.to(DEV).requires_grad_(True)

    m = GumbelSoftmax({'discrete_size': K, 'cuda': True, 'half': False})
    m.train()
    z, params = m(lg, noise=U.to(DEV))
    d = params['discrete']
    assert isinstance(d, LazyParams) and d.is_pending('z_hard') and d.is_pending('log_q_z')
    assert list(d) == ['z_hard', 'logits', 'log_q_z', 'tau_scalar']
    y_e, zh_e, lq_e = F_.gumbel_softmax(lg, U.to(DEV), tau=1.0)            # the eager launch: all three outputs
    assert torch.equal(z, y_e)
    assert torch.equal(d['z_hard'], zh_e) and torch.equal(d['log_q_z'], lq_e)
    y0, p0 = O.gumbel_forward(logits, U, tau=1.0, training=True)
    assert torch.equal((d['z_hard'].cpu() == 1), (p0['discrete']['z_hard'] == 1))
    w = torch.randn(B, K)
    lc = logits.clone().requires_grad_(True)
    y0, p0 = O.gumbel_forward(lc, U, tau=1.0, training=True)
    (g0,) = torch.autograd.grad((p0['discrete']['z_hard'] * w).sum() + (p0['discrete']['log_q_z'] * w).sum()
                                + (y0 * w).sum(), lc)
    wd = w.to(DEV)
    (g1,) = torch.autograd.grad((d['z_hard'] * wd).sum() + (d['log_q_z'] * wd).sum() + (z * wd).sum(), lg)
    close(g1, g0, rtol=1e-4, atol=1e-5)

    mb = Bernoulli({'discrete_size': K, 'cuda': True, 'half': False, 'epochs': 10})
    mb.train()
    zb, pb = mb(lg, noise=U.to(DEV))
    db = pb['discrete']
    assert isinstance(db, LazyParams) and db.is_pending('z_hard') and list(db) == ['z_hard', 'logits', 'tau_scalar']
    y_e, zh_e = F_.relaxed_bernoulli(lg, U.to(DEV), tau=1.0)
    assert torch.equal(zb, y_e) and torch.equal(db['z_hard'], zh_e)
    (gb,) = torch.autograd.grad((db['z_hard'] * wd).sum(), lg)
    (ge,) = torch.autograd.grad((zh_e * wd).sum(), lg)
    close(gb, ge, rtol=1e-5, atol=1e-7)
