"""The non-temporal store branch of every kernel family against the ORACLE.

Stores are non-temporal when one [B, row] tensor of the launch reaches 32 MiB (csrc/common.hpp: nt_stores); that
branch is an inline-asm `global_store_dwordx4 ... nt` which the compiler neither schedules nor pads for hazards
(store_vec).  A missing wait state or a clobbered data register there shows up as wrong VALUES in some rows, so
every family is driven through a launch that large and every row of every output -- and of the gradients -- is
compared with the CPU oracle (not with a small launch of the same kernel, which shares the plain-store path's
arithmetic but not the nt branch's hazards).  Sizes: 33.6-43 MB per tensor, ~10 M elements: the oracle takes a
second or two per family."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
NT = 32 << 20


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def big(*shape):
    assert shape[0] * shape[-1] * 4 >= NT or (len(shape) == 3 and shape[0] * shape[1] * shape[2] * 4 >= NT), shape
    return shape


@pytest.fixture(autouse=True)
def _free():
    yield
    torch.cuda.empty_cache()


def test_gauss_forward_backward_and_pair_kl():
    from vae_amd import functional as F_
    torch.manual_seed(1)
    B, L = big(16400, 512)
    logits, noise, w, wk = torch.randn(B, 2 * L), torch.randn(B, L), torch.randn(B, L), torch.randn(B)
    lc = logits.clone().requires_grad_(True)
    z0, p0 = O.gauss_forward(lc, noise)
    k0 = O.gauss_kl(p0)
    (g0,) = torch.autograd.grad((z0 * w).sum() + (k0 * wk).sum(), lc)
    ld = logits.to(DEV).requires_grad_(True)
    z1, lv1, k1, _ = F_.gauss_reparam_kl(ld, noise.to(DEV))
    (g1,) = torch.autograd.grad((z1 * w.to(DEV)).sum() + (k1 * wk.to(DEV)).sum(), ld)
    assert torch.equal(z1.detach().cpu(), z0.detach()) and torch.equal(lv1.cpu(), p0['gaussian']['logvar'].detach())
    close(k1, k0, rtol=1e-5, atol=1e-3)
    close(g1, g0, rtol=1e-4, atol=1e-5)
    # the stand-alone KL against a prior: four gradient tensors of [B, L]
    mq, sq, mp, sp = torch.randn(B, L), torch.rand(B, L) + 0.3, torch.randn(B, L), torch.rand(B, L) + 0.3
    c = [t.clone().requires_grad_(True) for t in (mq, sq, mp, sp)]
    kk0 = O.gauss_kl({'gaussian': {'mu': c[0], 'logvar': c[1]}}, {'gaussian': {'mu': c[2], 'logvar': c[3]}})
    gg0 = torch.autograd.grad((kk0 * wk).sum(), c)
    d = [t.to(DEV).requires_grad_(True) for t in (mq, sq, mp, sp)]
    kk1 = F_.gauss_kl(*d)
    gg1 = torch.autograd.grad((kk1 * wk.to(DEV)).sum(), d)
    close(kk1, kk0, rtol=1e-5, atol=1e-3)
    for a, b in zip(gg1, gg0):
        close(a, b, rtol=1e-4, atol=1e-5)


def test_gumbel_and_categorical_kl():
    from vae_amd import functional as F_
    torch.manual_seed(2)
    B, K = big(16400, 512)
    logits, U, w = torch.randn(B, K), torch.rand(B, K), torch.randn(B, K)
    lc = logits.clone().requires_grad_(True)
    y0, p0 = O.gumbel_forward(lc, U, tau=0.8, training=True)
    d0 = p0['discrete']
    (g0,) = torch.autograd.grad((y0 * w).sum() + (d0['log_q_z'] * w).sum() * 0.5 + (d0['z_hard'] * w).sum() * 0.25, lc)
    ld = logits.to(DEV).requires_grad_(True)
    y1, zh1, lq1 = F_.gumbel_softmax(ld, U.to(DEV), tau=0.8)
    (g1,) = torch.autograd.grad((y1 * w.to(DEV)).sum() + (lq1 * w.to(DEV)).sum() * 0.5 + (zh1 * w.to(DEV)).sum() * 0.25, ld)
    close(y1, y0, rtol=1e-5, atol=1e-7)
    close(lq1, d0['log_q_z'], rtol=1e-5, atol=1e-5)
    assert torch.equal(zh1.cpu() == 1, d0['z_hard'] == 1)                  # every hot position of 16400 x 512
    close(g1, g0, rtol=1e-4, atol=1e-6)
    a, b, wk = torch.log_softmax(torch.randn(B, K), -1), torch.log_softmax(torch.randn(B, K), -1), torch.randn(B)
    ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    k0 = O.categorical_kl({'discrete': {'log_q_z': ac}}, {'discrete': {'log_q_z': bc}})
    gg0 = torch.autograd.grad((k0 * wk).sum(), [ac, bc])
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    k1 = F_.categorical_kl(ad, bd)
    gg1 = torch.autograd.grad((k1 * wk.to(DEV)).sum(), [ad, bd])
    close(k1, k0, rtol=1e-5, atol=1e-5)
    close(gg1[0], gg0[0], rtol=1e-4, atol=1e-6)
    close(gg1[1], gg0[1], rtol=1e-4, atol=1e-6)


def test_bernoulli_sample_and_kl():
    from vae_amd import functional as F_
    torch.manual_seed(3)
    B, K = big(16400, 512)
    logits, U, w, wk = torch.randn(B, K) * 2, torch.rand(B, K), torch.randn(B, K), torch.randn(B)
    lc = logits.clone().requires_grad_(True)
    y0, p0 = O.bernoulli_forward(lc, U, tau=0.7, training=True)
    (g0,) = torch.autograd.grad((y0 * w).sum() + (p0['discrete']['z_hard'] * w).sum() * 0.5, lc)
    ld = logits.to(DEV).requires_grad_(True)
    y1, zh1 = F_.relaxed_bernoulli(ld, U.to(DEV), tau=0.7)
    (g1,) = torch.autograd.grad((y1 * w.to(DEV)).sum() + (zh1 * w.to(DEV)).sum() * 0.5, ld)
    close(y1, y0, rtol=1e-5, atol=1e-6)
    hot1, hot0 = zh1.cpu() == 1, p0['discrete']['z_hard'] == 1
    edge = (y0.detach() - 0.5).abs() < 1e-6
    assert torch.equal(hot1 | edge, hot0 | edge)
    close(g1, g0, rtol=1e-4, atol=1e-5)
    prior = torch.randn(B, K)
    qc, pc = logits.clone().requires_grad_(True), prior.clone().requires_grad_(True)
    k0 = O.bernoulli_kl({'discrete': {'logits': qc}}, {'discrete': {'logits': pc}})
    gg0 = torch.autograd.grad((k0 * wk).sum(), [qc, pc])
    qd, pd = logits.to(DEV).requires_grad_(True), prior.to(DEV).requires_grad_(True)
    k1 = F_.bernoulli_kl(qd, pd)
    gg1 = torch.autograd.grad((k1 * wk.to(DEV)).sum(), [qd, pd])
    close(k1, k0, rtol=1e-5, atol=1e-4)
    close(gg1[0], gg0[0], rtol=1e-4, atol=1e-5)
    close(gg1[1], gg0[1], rtol=1e-4, atol=1e-5)


def test_beta_split_and_kl():
    from vae_amd import functional as F_
    torch.manual_seed(4)
    B, L = big(16400, 512)
    logits, wk, w = torch.randn(B, 2 * L) * 2, torch.randn(B), torch.randn(B, L)
    # fp64 oracle for the gradient: digamma / trigamma differences are ill-conditioned in fp32 on both sides
    lc = logits.double().requires_grad_(True)
    p0 = O.beta_params(lc)
    k0 = O.beta_kl({'beta': p0})
    (g0,) = torch.autograd.grad((k0 * wk.double()).sum() + (p0['conc1'] * w.double()).sum() - (p0['conc2'] * w.double()).sum(), lc)
    p0 = {k: v.float() for k, v in p0.items()}
    k0, g0 = k0.float(), g0.float()
    ld = logits.to(DEV).requires_grad_(True)
    c1, c0, k1, _ = F_.beta_split_kl(ld)
    (g1,) = torch.autograd.grad((k1 * wk.to(DEV)).sum() + (c1 * w.to(DEV)).sum() - (c0 * w.to(DEV)).sum(), ld)
    close(c1, p0['conc1'], rtol=1e-6, atol=1e-7)
    close(c0, p0['conc2'], rtol=1e-6, atol=1e-7)
    close(k1, k0, rtol=1e-5, atol=1e-3)
    close(g1, g0, rtol=1e-3, atol=1e-4)
    a = [torch.rand(B, L) * 0.9 + 0.05 for _ in range(4)]
    ac = [t.clone().requires_grad_(True) for t in a]
    kk0 = O.beta_kl({'beta': {'conc1': ac[0], 'conc2': ac[1]}}, {'beta': {'conc1': ac[2], 'conc2': ac[3]}})
    gg0 = torch.autograd.grad((kk0 * wk).sum(), ac)
    ad = [t.to(DEV).requires_grad_(True) for t in a]
    kk1 = F_.beta_kl(*ad)
    gg1 = torch.autograd.grad((kk1 * wk.to(DEV)).sum(), ad)
    close(kk1, kk0, rtol=1e-5, atol=1e-3)
    for x1, x0 in zip(gg1, gg0):
        close(x1, x0, rtol=1e-4, atol=1e-4)


def test_flow_kl_backward():
    from vae_amd import functional as F_
    torch.manual_seed(5)
    B, Fd = big(16400, 512)
    u, ld, wk = torch.randn(B, Fd), torch.randn(B, 1), torch.randn(B)
    uc, lc = u.clone().requires_grad_(True), ld.clone().requires_grad_(True)
    k0 = O.flow_kl({'recon_x': uc, 'logdet': lc})
    g0 = torch.autograd.grad((k0 * wk).sum(), [uc, lc])
    ud, ldd = u.to(DEV).requires_grad_(True), ld.to(DEV).requires_grad_(True)
    k1 = F_.flow_kl(ud, ldd)
    g1 = torch.autograd.grad((k1 * wk.to(DEV)).sum(), [ud, ldd])
    close(k1, k0, rtol=1e-5, atol=1e-3)
    close(g1[0], g0[0], rtol=1e-4, atol=1e-7)
    close(g1[1], g0[1], rtol=1e-5, atol=1e-7)


def test_vrnn_pair_and_multi_sample_gauss():
    from vae_amd import functional as F_
    torch.manual_seed(6)
    B, L = big(16400, 512)
    enc, pri, n1, n2 = torch.randn(B, 2 * L), torch.randn(B, 2 * L), torch.randn(B, L), torch.randn(B, L)
    w1, w2, wk = torch.randn(B, L), torch.randn(B, L), torch.randn(B)
    ec, pc = enc.clone().requires_grad_(True), pri.clone().requires_grad_(True)
    z0, p0 = O.vrnn_reparameterize(ec, pc, n1, n2)
    k0 = O.vrnn_kld(p0, True)
    g0 = torch.autograd.grad((z0['posterior'] * w1).sum() + (z0['prior'] * w2).sum() + (k0 * wk).sum()
                             + (p0['prior']['logits'] * torch.cat([w2, w1], -1)).sum(), [ec, pc])
    ed, pd = enc.to(DEV).requires_grad_(True), pri.to(DEV).requires_grad_(True)
    zq, zp, _, _, clamped, klp, kl0, _ = F_.gauss_pair(ed, pd, n1.to(DEV), n2.to(DEV), with_prior_kl=True,
                                                        want_logvars=False)
    g1 = torch.autograd.grad((zq * w1.to(DEV)).sum() + (zp * w2.to(DEV)).sum() + ((klp + kl0) * wk.to(DEV)).sum()
                             + (clamped * torch.cat([w2, w1], -1).to(DEV)).sum(), [ed, pd])
    assert torch.equal(zq.cpu(), z0['posterior'])
    close(zp, z0['prior'], rtol=1e-6, atol=1e-6)
    close(clamped, p0['prior']['logits'], rtol=1e-6, atol=1e-7)
    close(klp + kl0, k0, rtol=1e-5, atol=1e-2)
    close(g1[0], g0[0], rtol=1e-4, atol=1e-4)
    close(g1[1], g0[1], rtol=1e-4, atol=1e-4)
    # S = 3 draws of the same logits: z [S, B, L]
    S = 3
    noises = [torch.randn(B, L) for _ in range(S)]
    ws = torch.randn(S, B, L)
    lc = enc.clone().requires_grad_(True)
    zs0, ps0 = O.gauss_forward_multi(lc, noises)
    (gm0,) = torch.autograd.grad(sum((z * ws[i]).sum() for i, z in enumerate(zs0)), lc)
    ld = enc.to(DEV).requires_grad_(True)
    zs1, _, _, _ = F_.gauss_reparam_multi(ld, S, noise=torch.stack(noises, 0).to(DEV))
    (gm1,) = torch.autograd.grad((zs1 * ws.to(DEV)).sum(), ld)
    for i in range(S):
        assert torch.equal(zs1[i].cpu(), zs0[i])
    close(gm1, gm0, rtol=1e-4, atol=1e-5)


def close_cond(a, ref32, ref64, rtol=1e-5, atol=1e-6):
    """tests/test_gpu_elbo.py: for functions that are ill-conditioned in fp32 (the variance NLLs' gradients, the
    discretized logistic's value) the kernel must be about as accurate against an fp64 evaluation as the fp32 oracle
    is: L2 and max error within 2x the oracle's, plus rtol / atol of the fp64 value."""
    a = a.detach().double().cpu()
    ref32, ref64 = ref32.detach().double(), ref64.detach().double()
    e_k, e_o = (a - ref64), (ref32 - ref64)
    assert float(e_k.norm()) <= 2 * float(e_o.norm()) + rtol * float(ref64.norm()) + atol
    assert float(e_k.abs().max()) <= 2 * float(e_o.abs().max()) + rtol * float(ref64.abs().max()) + atol


@pytest.mark.parametrize('nll_type', ['bernoulli', 'gaussian', 'disc_logistic'])
def test_nll_gradients(nll_type):
    """The general backward, the one-pass forward+gradient and the multi-sample backward: d recon for every row."""
    from vae_amd import functional as F_
    torch.manual_seed(7)
    B, Dp = big(11000, 784)
    R = Dp if nll_type == 'bernoulli' else 2 * Dp
    x = torch.floor(torch.rand(B, Dp) * 256) / 255.0 if nll_type != 'bernoulli' else (torch.rand(B, Dp) < 0.3).float()
    x = x.clamp(0, 1)
    recon, recon_b, kld, wk = torch.randn(B, R), torch.randn(B, R), torch.rand(B), torch.randn(B)

    def oracle(dt):
        rc, rb = recon.clone().to(dt).requires_grad_(True), recon_b.clone().to(dt).requires_grad_(True)
        n = O.nll(x.to(dt), rc, nll_type)
        out = O.loss_function(n, kld.to(dt), 0.7)
        (g,) = torch.autograd.grad(out['loss_mean'] + (out['loss'] * wk.to(dt)).sum() / B, rc, retain_graph=True)
        (gm,) = torch.autograd.grad(out['loss_mean'], rc)
        nm = O.nll_multi(x.to(dt), [rc, rb], nll_type)
        gab = torch.autograd.grad((nm * wk.to(dt)).sum() / B, [rc, rb])
        return n.detach(), g, gm, nm.detach(), gab
    n32, g32, gm32, nm32, gab32 = oracle(torch.float32)
    n64, g64, gm64, nm64, gab64 = oracle(torch.float64)

    def check(v, v32, v64, value=False):
        if nll_type == 'bernoulli':
            close(v, v32, rtol=1e-5, atol=1e-3 if value else 2e-8)
        else:
            close_cond(v, v32, v64)
    rd = recon.to(DEV).requires_grad_(True)
    n1, l1, e1, m1 = F_.nll_elbo(x.to(DEV), rd, nll_type, kld=kld.to(DEV), kl_beta=0.7)
    (g1,) = torch.autograd.grad(m1[0] + (l1 * wk.to(DEV)).sum() / B, rd)
    check(n1, n32, n64, value=True)
    check(g1, g32, g64)
    # one pass: loss_mean.backward() alone
    rd2 = recon.to(DEV).requires_grad_(True)
    _, _, _, m2 = F_.nll_elbo(x.to(DEV), rd2, nll_type, kld=kld.to(DEV), kl_beta=0.7, fuse_backward=True)
    m2[0].backward()
    check(rd2.grad, gm32, gm64)
    # two decoder outputs, x read once
    rda, rdb = recon.to(DEV).requires_grad_(True), recon_b.to(DEV).requires_grad_(True)
    nm1, _, _, _ = F_.nll_multi_elbo(x.to(DEV), [rda, rdb], nll_type)
    gm1 = torch.autograd.grad((nm1 * wk.to(DEV)).sum() / B, [rda, rdb])
    check(nm1, nm32, nm64, value=True)
    check(gm1[0], gab32[0], gab64[0])
    check(gm1[1], gab32[1], gab64[1])


def test_flow_tails():
    from vae_amd import functional as F_
    torch.manual_seed(8)
    B, Fd = big(16400, 512)
    x, m, a, wu, wl = torch.randn(B, Fd), torch.randn(B, Fd), torch.randn(B, Fd) * 0.5, torch.randn(B, Fd), torch.randn(B)
    for tanh in (False, True):
        c = [t.clone().requires_grad_(True) for t in (x, m, a)]
        u0, ld0 = O.made_tail(*c, pre_exp_tanh=tanh)
        g0 = torch.autograd.grad((u0 * wu).sum() + (ld0.view(-1) * wl).sum(), c)
        d = [t.to(DEV).requires_grad_(True) for t in (x, m, a)]
        u1, ld1 = F_.made_tail(*d, pre_exp_tanh=tanh)
        g1 = torch.autograd.grad((u1 * wu.to(DEV)).sum() + (ld1 * wl.to(DEV)).sum(), d)
        close(u1, u0, rtol=1e-5, atol=1e-6)
        close(ld1, ld0.view(-1), rtol=1e-5, atol=1e-4)
        for p, q in zip(g1, g0):
            close(p, q, rtol=1e-4, atol=1e-5)
    mask = (torch.arange(Fd) % 2).float()
    c = [t.clone().requires_grad_(True) for t in (x, a, m)]
    y0, ld0 = O.coupling_tail(c[0], c[1], c[2], mask)
    g0 = torch.autograd.grad((y0 * wu).sum() + (ld0.view(-1) * wl).sum(), c)
    d = [t.to(DEV).requires_grad_(True) for t in (x, a, m)]
    y1, ld1 = F_.coupling_tail(d[0], d[1], d[2], mask.to(DEV))
    g1 = torch.autograd.grad((y1 * wu.to(DEV)).sum() + (ld1 * wl.to(DEV)).sum(), d)
    close(y1, y0, rtol=1e-5, atol=1e-6)
    close(ld1, ld0.view(-1), rtol=1e-5, atol=1e-4)
    for p, q in zip(g1, g0):
        close(p, q, rtol=1e-4, atol=1e-5)
    # BatchNormFlow (training: batch statistics) with the Reverse folded into the store
    lg, beta = torch.randn(Fd) * 0.3, torch.randn(Fd)
    xs = x * 1.7 + 0.4
    c = [t.clone().requires_grad_(True) for t in (xs, lg, beta)]
    y0, ld0, mean0, var0 = O.bnflow(c[0], c[1], c[2], None, None, training=True)
    y0 = O.reverse_columns(y0)
    g0 = torch.autograd.grad((y0 * wu).sum() + ld0.sum() * wl.sum(), c)
    d = [t.to(DEV).requires_grad_(True) for t in (xs, lg, beta)]
    y1, ld1, mean1, var1 = F_.bnflow(d[0], d[1], d[2], training=True, reverse=True)
    g1 = torch.autograd.grad((y1 * wu.to(DEV)).sum() + (ld1 * wl.to(DEV)).sum(), d)
    close(y1, y0, rtol=1e-4, atol=1e-5)
    close(mean1, mean0, rtol=1e-5, atol=1e-6)
    close(var1, var0, rtol=1e-5, atol=1e-6)
    close(g1[0], g0[0], rtol=1e-3, atol=1e-5)
    close(g1[1], g0[1], rtol=1e-3, atol=1e-2)
    close(g1[2], g0[2], rtol=1e-4, atol=1e-3)


def test_concat_row_kernel():
    """Concat(Gaussian, Gumbel) through the one-launch row kernel, forward and backward, against the oracle's
    per-child restatement."""
    from vae_amd.reparameterizers import ConcatReparameterizer
    torch.manual_seed(9)
    B, L, K = 22000, 128, 256        # the [B, L + K] sample row: 33.8 MB
    assert B * (L + K) * 4 >= NT
    cfg = {'continuous_size': 2 * L, 'discrete_size': K, 'cuda': True, 'half': False, 'epochs': 10}
    for train in (True, False):
        m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'], dict(cfg)).train(train)
        logits, noise, U, w = torch.randn(B, 2 * L + K), torch.randn(B, L), torch.rand(B, K), torch.randn(B, L + K)
        lc = logits.clone().requires_grad_(True)
        z0, p0 = O.concat_forward(lc, [('isotropic_gaussian', 2 * L, {}), ('discrete', K, {'tau': 1.0})],
                                  [noise if train else None, U], training=train)
        k0 = O.concat_kl(p0, ['isotropic_gaussian', 'discrete'])
        (g0,) = torch.autograd.grad((z0 * w).sum() + k0.sum(), lc)
        ld = logits.to(DEV).requires_grad_(True)
        z1, p1 = m(ld, noise=[noise.to(DEV) if train else None, U.to(DEV)])
        k1 = m.kl(p1)
        (g1,) = torch.autograd.grad((z1 * w.to(DEV)).sum() + k1.sum(), ld)
        assert torch.equal(z1[:, :L].cpu(), z0[:, :L].detach())
        if train:
            close(z1[:, L:], z0[:, L:], rtol=1e-5, atol=1e-7)
        else:
            assert torch.equal(z1[:, L:].cpu() == 1, z0[:, L:] == 1)
        close(k1, k0, rtol=1e-5, atol=1e-3)
        close(g1, g0, rtol=1e-4, atol=1e-5)


def test_made_tail_with_moments_then_bnflow():
    """The MADE tail that also leaves the BatchNormFlow moments (its u store takes the nt branch) followed by the
    moments-given BatchNormFlow forward + Reverse, against the oracle's two layers."""
    from vae_amd import functional as F_
    torch.manual_seed(10)
    B, Fd = big(16400, 512)
    x, ma = torch.randn(B, Fd) * 1.5 + 0.7, torch.randn(B, 2 * Fd) * 0.6
    lg, beta = torch.randn(Fd) * 0.3, torch.randn(Fd)
    wu, wl = torch.randn(B, Fd), torch.randn(B)
    c = [t.clone().requires_grad_(True) for t in (x, ma, lg, beta)]
    u0, ld0 = O.made_tail(c[0], c[1][:, :Fd], c[1][:, Fd:])
    y0, lb0, mean0, var0 = O.bnflow(u0, c[2], c[3], None, None, training=True)
    y0 = O.reverse_columns(y0)
    l0 = ld0.view(-1) + lb0
    g0 = torch.autograd.grad((y0 * wu).sum() + (l0 * wl).sum(), c)
    d = [t.to(DEV).requires_grad_(True) for t in (x, ma, lg, beta)]
    stats = F_.bn_stats_buffer(d[0])
    assert stats is not None
    u1, ld1 = F_.made_tail(d[0], d[1], None, bn_stats=stats)
    y1, l1, mean1, var1 = F_.bnflow(u1, d[2], d[3], logdet_in=ld1, training=True, reverse=True, moments=stats)
    g1 = torch.autograd.grad((y1 * wu.to(DEV)).sum() + (l1 * wl.to(DEV)).sum(), d)
    close(u1, u0, rtol=1e-5, atol=1e-6)
    close(mean1, mean0, rtol=1e-5, atol=1e-6)
    close(var1, var0, rtol=2e-5, atol=1e-6)
    close(y1, y0, rtol=1e-4, atol=2e-5)
    close(l1, l0, rtol=1e-5, atol=1e-3)
    close(g1[0], g0[0], rtol=1e-3, atol=1e-5)
    close(g1[1], g0[1], rtol=1e-3, atol=1e-5)
    close(g1[2], g0[2], rtol=1e-3, atol=2e-2)
    close(g1[3], g0[3], rtol=1e-4, atol=1e-3)


def test_gauss_4d_logits():
    """4-D logits read where they are (channel halves, KL over W): [B, C, H, W] with the sample tensor >= 32 MiB."""
    from vae_amd.reparameterizers import IsotropicGaussian
    torch.manual_seed(11)
    B, C, H, W = 520, 64, 16, 32         # z: [520, 32, 16, 32] = 34 MB
    assert B * (C // 2) * H * W * 4 >= NT
    logits, noise = torch.randn(B, C, H, W), torch.randn(B, C // 2, H, W)
    w = torch.randn(B, C // 2, H, W)
    lc = logits.clone().requires_grad_(True)
    z0, p0 = O.gauss_forward(lc, noise)
    k0 = O.gauss_kl(p0)
    (g0,) = torch.autograd.grad((z0 * w).sum() + k0.sum(), lc)
    m = IsotropicGaussian({'continuous_size': C, 'cuda': True, 'half': False}).train()
    ld = logits.to(DEV).requires_grad_(True)
    z1, p1 = m(ld, noise=noise.to(DEV))
    k1 = m.kl(p1)
    (g1,) = torch.autograd.grad((z1 * w.to(DEV)).sum() + k1.sum(), ld)
    assert z1.shape == z0.shape and torch.equal(z1.detach().cpu(), z0.detach())
    close(k1, k0, rtol=1e-5, atol=1e-4)
    close(g1, g0, rtol=1e-4, atol=1e-5)
