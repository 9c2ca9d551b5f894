"""16-bit tensors -- bf16 (north_star: fp32 + bf16; SURVEY 8c: bf16 rtol 1e-2) and fp16 (the reference's `half=True`
configs, helpers.utils.same_type) -- through every kernel family.

The kernels read bf16, compute in fp32 and round once on the way out, so each result is compared with the
oracle evaluated in fp32 on the SAME bf16-representable inputs: |a - b| <= 1e-2 |b| + 1e-2 max|b| (one bf16 ulp
is 2^-8 = 0.4 % of the value; the absolute term covers entries that are small next to the largest of the tensor)."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BF = torch.bfloat16
HALVES = [torch.bfloat16, torch.float16]


def close16(a, b, rtol=1e-2, afrac=1e-2):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    atol = afrac * max(float(b.abs().max()), 1e-6) if b.numel() else 0.0
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


_DT = [torch.bfloat16]


@pytest.fixture(autouse=True, params=HALVES, ids=['bf16', 'fp16'])
def _dtype(request):
    _DT[0] = request.param
    yield


def bf(*shape, seed=0, scale=1.0, rand=False):
    g = torch.Generator().manual_seed(seed)
    t = torch.rand(*shape, generator=g) if rand else torch.randn(*shape, generator=g) * scale
    return t.to(_DT[0])


@pytest.mark.parametrize('B,K', [(5, 3), (64, 32), (300, 256), (33, 1024)])
def test_gumbel_bf16(B, K):
    from vae_amd import functional as F_
    logits, U = bf(B, K, seed=1), bf(B, K, seed=2, rand=True)
    # the sample path mirrors the reference's per-op roundings in the tensor dtype (gumbel.py:269-273 casts the
    # noise to the logits dtype, then every op rounds): the matching oracle is the restatement run IN bf16
    lc = logits.clone().requires_grad_(True)
    # (the noise transform itself runs in fp32 on the fp32 CPU draw, gumbel.py:269-271, and only g is cast)
    y0, p0 = O.gumbel_forward(lc, U.float(), tau=0.8, training=True)
    lq0 = p0['discrete']['log_q_z']
    lg = logits.to(DEV).requires_grad_(True)
    y1, h1, lq1 = F_.gumbel_softmax(lg, U.to(DEV), tau=0.8)
    assert y1.dtype == _DT[0] and h1.dtype == _DT[0] and lq1.dtype == _DT[0]
    close16(y1, y0, rtol=2e-2, afrac=1e-2)
    close16(lq1, lq0)
    hot = h1.detach().float().cpu() == 1
    assert int(hot.sum(-1).min()) >= 1
    # every hot entry is a row maximum of the kernel's own stored soft sample
    ys = y1.detach().float().cpu()
    assert bool((ys[hot] == ys.max(-1, keepdim=True)[0].expand_as(ys)[hot]).all())
    w, v = bf(B, K, seed=3), bf(B, K, seed=4)
    (g0,) = torch.autograd.grad((y0.float() * w.float()).sum() + (lq0.float() * v.float()).sum(), lc)
    (g1,) = torch.autograd.grad((y1.float() * w.float().to(DEV)).sum() + (lq1.float() * v.float().to(DEV)).sum(), lg)
    assert g1.dtype == _DT[0]
    close16(g1, g0, rtol=5e-2, afrac=3e-2)


@pytest.mark.parametrize('B,K', [(7, 5), (128, 256)])
def test_categorical_kl_bf16(B, K):
    from vae_amd import functional as F_
    q, p = bf(B, K, seed=5), bf(B, K, seed=6)
    qc, pc = q.float().requires_grad_(True), p.float().requires_grad_(True)
    k0 = O.categorical_kl({'discrete': {'log_q_z': qc}}, {'discrete': {'log_q_z': pc}})
    qd, pd = q.to(DEV).requires_grad_(True), p.to(DEV).requires_grad_(True)
    k1 = F_.categorical_kl(qd, pd)
    close16(k1, k0)
    g0 = torch.autograd.grad(k0.sum(), [qc, pc])
    g1 = torch.autograd.grad(k1.float().sum(), [qd, pd])
    for a, b in zip(g1, g0):
        close16(a, b, rtol=2e-2, afrac=2e-2)


@pytest.mark.parametrize('B,K', [(5, 3), (200, 64), (17, 1024)])
def test_bernoulli_bf16(B, K):
    from vae_amd import functional as F_
    logits, U = bf(B, K, seed=7, scale=2.0), bf(B, K, seed=8, rand=True)
    # as for Gumbel: clamp_probs uses the dtype's own eps (torch/distributions/utils.py:101-124) and every op
    # rounds in bf16, so the oracle runs in bf16 too
    lc = logits.clone().requires_grad_(True)
    y0, p0 = O.bernoulli_forward(lc, U, tau=0.7, training=True)
    lg = logits.to(DEV).requires_grad_(True)
    y1, h1 = F_.relaxed_bernoulli(lg, U.to(DEV), tau=0.7)
    assert y1.dtype == _DT[0]
    close16(y1, y0, rtol=2e-2, afrac=1e-2)
    # hard sample: the threshold applied to the kernel's own stored relaxed value
    ys = y1.detach().float().cpu()
    assert torch.equal(h1.detach().float().cpu() == 1, ys >= 0.5)
    w = bf(B, K, seed=9)
    (g0,) = torch.autograd.grad((y0.float() * w.float()).sum(), lc)
    (g1,) = torch.autograd.grad((y1.float() * w.float().to(DEV)).sum(), lg)
    close16(g1, g0, rtol=5e-2, afrac=3e-2)
    for prior in (None, bf(B, K, seed=10)):
        k0 = O.bernoulli_kl({'discrete': {'logits': logits.float()}},
                            None if prior is None else {'discrete': {'logits': prior.float()}})
        k1 = F_.bernoulli_kl(logits.to(DEV), None if prior is None else prior.to(DEV))
        close16(k1, k0)


@pytest.mark.parametrize('B,L', [(6, 4), (129, 128)])
def test_beta_bf16(B, L):
    from vae_amd import functional as F_
    logits = bf(B, 2 * L, seed=11, scale=1.5)
    lc = logits.float().requires_grad_(True)
    p0 = O.beta_params(lc)
    k0 = O.beta_kl({'beta': p0})
    lg = logits.to(DEV).requires_grad_(True)
    c1, c0, k1, means = F_.beta_split_kl(lg)
    assert c1.dtype == _DT[0]
    close16(c1, p0['conc1'])
    close16(c0, p0['conc2'])
    close16(k1, k0)
    close16(means[0], p0['conc1_mean'])
    (g0,) = torch.autograd.grad(k0.sum(), lc)
    (g1,) = torch.autograd.grad(k1.float().sum(), lg)
    close16(g1, g0, rtol=2e-2, afrac=2e-2)
    # explicit pair of Betas
    a1, a0 = (bf(B, L, seed=12, rand=True).float() + 0.3).to(_DT[0]), (bf(B, L, seed=13, rand=True).float() + 0.3).to(_DT[0])
    b1, b0 = (bf(B, L, seed=14, rand=True).float() + 0.5).to(_DT[0]), (bf(B, L, seed=15, rand=True).float() + 0.5).to(_DT[0])
    kk0 = O.beta_kl({'beta': {'conc1': a1.float(), 'conc2': a0.float()}},
                    {'beta': {'conc1': b1.float(), 'conc2': b0.float()}})
    kk1 = F_.beta_kl(a1.to(DEV), a0.to(DEV), b1.to(DEV), b0.to(DEV))
    close16(kk1, kk0)


@pytest.mark.parametrize('B,F', [(9, 6), (256, 512)])
def test_flow_kl_bf16(B, F):
    from vae_amd import functional as F_
    u, ld = bf(B, F, seed=16), bf(B, 1, seed=17)
    uc, lc = u.float().requires_grad_(True), ld.float().requires_grad_(True)
    k0 = O.flow_kl({'recon_x': uc, 'logdet': lc})
    ud, ldd = u.to(DEV).requires_grad_(True), ld.to(DEV).requires_grad_(True)
    k1 = F_.flow_kl(ud, ldd)
    close16(k1, k0)
    w = bf(B, seed=18).float()
    g0 = torch.autograd.grad((k0 * w).sum(), [uc, lc])
    g1 = torch.autograd.grad((k1.float() * w.to(DEV)).sum(), [ud, ldd])
    for a, b in zip(g1, g0):
        close16(a, b.reshape(a.shape), rtol=2e-2, afrac=2e-2)


@pytest.mark.parametrize('B,L', [(5, 3), (96, 128)])
def test_gauss_pair_bf16(B, L):
    from vae_amd import functional as F_
    enc, pri = bf(B, 2 * L, seed=19), bf(B, 2 * L, seed=20)
    nq, npr = bf(B, L, seed=21), bf(B, L, seed=22)
    ec, pc = enc.float().requires_grad_(True), pri.float().requires_grad_(True)
    z0, p0 = O.vrnn_reparameterize(ec, pc, nq.float(), npr.float())
    kl0 = O.vrnn_kld(p0, use_prior_kl=True)
    ed, pd = enc.to(DEV).requires_grad_(True), pri.to(DEV).requires_grad_(True)
    zq, zp, lvq, lvp, clamped, klpair, klprior, means = F_.gauss_pair(ed, pd, nq.to(DEV), npr.to(DEV),
                                                                      with_prior_kl=True)
    assert zq.dtype == _DT[0]
    close16(zq, z0['posterior'])
    close16(zp, z0['prior'])
    # the pairwise KL divides by the clamped prior scale: compare at bf16 granularity of its largest terms
    close16(klpair.float() + klprior.float(), kl0, rtol=2e-2, afrac=2e-2)
    w = bf(B, L, seed=23).float()
    g0 = torch.autograd.grad((z0['posterior'] * w).sum() + kl0.sum(), [ec, pc])
    g1 = torch.autograd.grad((zq.float() * w.to(DEV)).sum() + klpair.float().sum() + klprior.float().sum(), [ed, pd])
    for a, b in zip(g1, g0):
        close16(a, b, rtol=2e-2, afrac=2e-2)


@pytest.mark.parametrize('B,L', [(4, 2), (130, 64), (64, 512)])
def test_gauss_backward_and_pair_kl_bf16(B, L):
    from vae_amd import functional as F_
    logits, noise = bf(B, 2 * L, seed=24), bf(B, L, seed=25)
    lc = logits.float().requires_grad_(True)
    z0, p0 = O.gauss_forward(lc, noise.float())
    k0 = O.gauss_kl(p0)
    lg = logits.to(DEV).requires_grad_(True)
    z1, lv1, k1, means = F_.gauss_reparam_kl(lg, noise.to(DEV))
    w = bf(B, L, seed=26).float()
    (g0,) = torch.autograd.grad((z0 * w).sum() + k0.sum(), lc)
    (g1,) = torch.autograd.grad((z1.float() * w.to(DEV)).sum() + k1.float().sum(), lg)
    assert g1.dtype == _DT[0]
    if _DT[0] == torch.float16:      # d KL / d s = s - 1/s leaves fp16's range for |s| < 1.5e-5: inf there, as in torch
        big = g0.abs() > 6e4
        assert bool(torch.isinf(g1.detach().cpu()[big]).all())
        g1, g0 = g1.detach().cpu().float().masked_fill(big, 0), g0.masked_fill(big, 0)
    close16(g1, g0, rtol=2e-2, afrac=2e-2)
    # KL between two explicit Gaussians (positive scales)
    mq, mp = bf(B, L, seed=27), bf(B, L, seed=28)
    sq = (bf(B, L, seed=29, rand=True).float() + 0.5).to(_DT[0])
    sp = (bf(B, L, seed=30, rand=True).float() + 0.5).to(_DT[0])
    kk0 = O.gauss_kl({'gaussian': {'mu': mq.float(), 'logvar': sq.float()}},
                     {'gaussian': {'mu': mp.float(), 'logvar': sp.float()}})
    kk1 = F_.gauss_kl(mq.to(DEV), sq.to(DEV), mp.to(DEV), sp.to(DEV))
    close16(kk1, kk0)
