"""GPU parity at BASELINE.json's full sizes (configs[1]..[4]) -- direct comparison with the oracle
where it finishes in seconds on the host, plus size-independent properties of the path:

  * shard invariance: rows [B/2, B) computed alone equal the same rows of the full launch bit for bit
    (every function on the path is per-sample independent);
  * elbo == nll + kld and loss == (nll + beta*kld) as single float operations (abstract_vae.py:356-361);
  * backward linearity: scaling the upstream gradient by a power of two scales the result exactly;
  * the batch means equal the double-precision mean of the per-sample outputs (rtol 1e-6).

Tolerances (north_star): z / logvar / hard masks bit-exact; KL / NLL / ELBO rtol 1e-5 fp32.
"""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def test_c2_gaussian_65536x512():
    """configs[1]: IsotropicGaussian latent 512, B = 65536 (logits 268 MB)."""
    from vae_amd.reparameterizers import IsotropicGaussian
    B, L = 65536, 512
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(B, 2 * L, generator=g)
    noise = torch.randn(B, L, generator=g)
    gz = torch.randn(B, L, generator=g)
    lc = logits.clone().requires_grad_(True)
    z0, p0 = O.gauss_forward(lc, noise)
    k0 = O.gauss_kl(p0)
    (g0,) = torch.autograd.grad((z0 * gz).sum() + k0.sum(), lc)

    m = IsotropicGaussian({'continuous_size': 2 * L, 'cuda': True, 'half': False})
    m.train()
    lg = logits.to(DEV).requires_grad_(True)
    nd = noise.to(DEV)
    z1, p1 = m(lg, noise=nd)
    k1 = m.kl(p1)
    assert torch.equal(z1.detach().cpu(), z0.detach())
    assert torch.equal(p1['gaussian']['logvar'].detach().cpu(), p0['gaussian']['logvar'].detach())
    close(k1, k0, rtol=1e-5, atol=1e-5 * L)
    close(p1['gaussian']['mu_mean'], p0['gaussian']['mu_mean'], atol=1e-6)
    close(p1['gaussian']['logvar_mean'], p0['gaussian']['logvar_mean'], atol=1e-6)
    gzd = gz.to(DEV)
    (g1,) = torch.autograd.grad((z1 * gzd).sum() + k1.sum(), lg, retain_graph=True)
    # d KL / d s = s - 1/s is unbounded near s = 0: rtol 1e-5 plus an absolute floor for the few rows there
    close(g1, g0, rtol=1e-5, atol=1e-5)

    # shard invariance (injected noise): second half alone == second half of the full launch
    zh, ph = m(lg[B // 2:], noise=nd[B // 2:])
    assert torch.equal(zh, z1[B // 2:])
    assert torch.equal(m.kl(ph), k1[B // 2:])
    # the mean equals the double mean of what the kernel read
    close(p1['gaussian']['mu_mean'], logits[:, :L].double().mean().float(), rtol=1e-6, atol=1e-7)
    # backward linearity under exact scaling
    (g2,) = torch.autograd.grad((z1 * (gzd * 4)).sum() + (k1 * 4).sum(), lg)
    assert torch.equal(g2, g1 * 4)


def test_c2_bernoulli_nll_elbo_65536x784():
    """configs[1]: 1x28x28 Bernoulli NLL + the ELBO combine at B = 65536."""
    from vae_amd import functional as F_
    from vae_amd.elbo import loss_function
    B, D = 65536, 784
    g = torch.Generator().manual_seed(12)
    x = (torch.rand(B, 1, 28, 28, generator=g) < 0.13).float()
    recon = torch.randn(B, 1, 28, 28, generator=g) * 2
    kld = torch.rand(B, generator=g) * 30
    rc = recon.clone().requires_grad_(True)
    n0 = O.nll(x, rc, 'bernoulli')
    out0 = O.loss_function(n0, kld, kl_beta=0.5)
    (g0,) = torch.autograd.grad(out0['loss_mean'], rc)

    xd, kd = x.to(DEV), kld.to(DEV)
    rd = recon.to(DEV).requires_grad_(True)
    out1 = loss_function(rd, xd, kd, 'bernoulli', kl_beta=0.5)
    n1 = F_.nll(xd, rd, 'bernoulli')
    close(n1, n0, rtol=1e-5, atol=1e-3)
    for k in ('loss', 'elbo'):
        close(out1[k], out0[k], rtol=1e-5, atol=1e-3)
    for k in ('loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'additional_loss_mean', 'mut_info_mean'):
        close(out1[k], out0[k], rtol=1e-5, atol=1e-5)
    (g1,) = torch.autograd.grad(out1['loss_mean'], rd)
    close(g1, g0, rtol=1e-5, atol=1e-10)
    # the combine is one float op per output on the launch's own nll (both the one-pass forward+gradient kernel
    # and the plain forward kernel)
    for fuse in (True, False):
        nn_, ll_, ee_, _ = F_.nll_elbo(xd, rd, 'bernoulli', kld=kd, kl_beta=0.5, fuse_backward=fuse)
        assert torch.equal(ee_, nn_.detach() + kd)
        assert torch.equal(ll_, nn_.detach() + 0.5 * kd)
        close(nn_, n0, rtol=1e-5, atol=1e-3)
    # means vs a double sum of the per-sample outputs
    close(out1['loss_mean'], out1['loss'].double().mean().float(), rtol=1e-6, atol=0)
    # shard invariance
    assert torch.equal(F_.nll(xd[B // 2:], rd[B // 2:], 'bernoulli'), n1[B // 2:])


def test_c3_gumbel_hard_65536x1024():
    """configs[2]: straight-through hard sample, K = 1024, B = 65536: every mask bit and index."""
    from vae_amd import functional as F_
    B, K = 65536, 1024
    g = torch.Generator().manual_seed(13)
    logits = torch.randn(B, K, generator=g)
    U = torch.rand(B, K, generator=g)
    y0, p0 = O.gumbel_forward(logits, U, tau=0.75, training=True)
    h0 = p0['discrete']['z_hard']
    y1, h1, lq1 = F_.gumbel_softmax(logits.to(DEV), U.to(DEV), tau=0.75)
    mask0, mask1 = h0 == 1, h1.cpu() == 1
    assert torch.equal(mask1, mask0)                                  # bit-exact hard sample (ties included)
    assert torch.equal(mask1.float().argmax(-1), mask0.float().argmax(-1))
    close(y1, y0, rtol=1e-5, atol=1e-9)                               # soft sample: exp/sum order may differ by ulps
    close(lq1, p0['discrete']['log_q_z'], rtol=1e-5, atol=1e-5)
    assert int(mask1.sum(-1).min()) >= 1


def test_c3_concat_gauss_gumbel_65536():
    """configs[2]: Concat(IsotropicGaussian L=128, GumbelSoftmax K=256) at B = 65536, one shared logits buffer."""
    from vae_amd.reparameterizers import ConcatReparameterizer
    B, L, K = 65536, 128, 256
    g = torch.Generator().manual_seed(14)
    logits = torch.randn(B, 2 * L + K, generator=g)
    noise = torch.randn(B, L, generator=g)
    U = torch.rand(B, K, generator=g)
    z0, p0 = O.concat_forward(logits, [('isotropic_gaussian', 2 * L, {}), ('discrete', K, {'tau': 1.0})], [noise, U],
                              training=True)
    k0 = O.concat_kl(p0, ['isotropic_gaussian', 'discrete'])
    m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'],
                              {'continuous_size': 2 * L, 'discrete_size': K, 'cuda': True, 'half': False})
    m.train()
    z1, p1 = m(logits.to(DEV), noise=[noise.to(DEV), U.to(DEV)])
    assert z1.shape == (B, L + K)
    assert torch.equal(z1[:, :L].cpu(), z0[:, :L])                 # Gaussian child: bit-exact
    close(z1[:, L:], z0[:, L:], rtol=1e-5, atol=1e-9)              # Gumbel soft sample
    h1 = p1['concat'][1]['discrete']['z_hard'].cpu() == 1
    assert torch.equal(h1, p0['concat'][1]['discrete']['z_hard'] == 1)   # hard mask: bit-exact
    close(m.kl(p1), k0, rtol=1e-5, atol=1e-5 * L)


@pytest.mark.parametrize('nll_type', ['disc_logistic', 'gaussian'])
def test_c4_variance_nll_3x64x64(nll_type):
    """configs[3]: 3x64x64 images, decoder rows [mean | variance] (D = 12288).  Oracle at B = 2048 (300 MB of
    decoder output); shard invariance at B = 8192."""
    from vae_amd import functional as F_
    B, C, H, W = 2048, 3, 64, 64
    g = torch.Generator().manual_seed(15)
    x = torch.floor(torch.rand(B, C, H, W, generator=g) * 256) / 255.0
    x = x.clamp(0, 1)
    recon = torch.randn(B, 2 * C, H, W, generator=g)
    n0 = O.nll(x, recon, nll_type)
    n1 = F_.nll(x.to(DEV), recon.to(DEV), nll_type)
    close(n1, n0, rtol=1e-5, atol=1e-2)
    Bb = 8192
    gd = torch.Generator(device=DEV).manual_seed(16)
    xb = torch.floor(torch.rand(Bb, C, H, W, generator=gd, device=DEV) * 256) / 255.0
    rb = torch.randn(Bb, 2 * C, H, W, generator=gd, device=DEV)
    full = F_.nll(xb, rb, nll_type)
    assert torch.equal(F_.nll(xb[Bb // 2:], rb[Bb // 2:], nll_type), full[Bb // 2:])
    assert bool(torch.isfinite(full).all())


def test_c5_vrnn_step_16384x128():
    """configs[4]: one VRNN time step at the per-GPU batch (posterior + clamped prior draw, both KLs)."""
    from vae_amd import functional as F_
    B, L = 16384, 128
    g = torch.Generator().manual_seed(17)
    enc = torch.randn(B, 2 * L, generator=g)
    pri = torch.randn(B, 2 * L, generator=g)
    nq, npr = torch.randn(B, L, generator=g), torch.randn(B, L, generator=g)
    z0, p0 = O.vrnn_reparameterize(enc, pri, nq, npr)
    kl0 = O.vrnn_kld(p0, use_prior_kl=True)
    zq, zp, lvq, lvp, clamped, klpair, klprior, means = F_.gauss_pair(
        enc.to(DEV), pri.to(DEV), nq.to(DEV), npr.to(DEV), with_prior_kl=True)
    assert torch.equal(zq.cpu(), z0['posterior'])
    close(zp, z0['prior'], rtol=1e-6, atol=1e-6)     # sigmoid in the clamp: 1-ulp forms differ
    close(klpair + klprior, kl0, rtol=1e-5, atol=1e-5 * L)


def test_beyond_int32_elements():
    """Tensors with more than 2^31 elements (64-bit row offsets everywhere): the last rows of a 9-10 GB launch equal
    the same rows computed alone, forward and backward, for the Gaussian pair of kernels and the one-pass NLL."""
    from vae_amd import functional as F_
    gd = torch.Generator(device=DEV).manual_seed(21)
    # Gaussian: B * 2L = 2.25e9 elements of logits
    B, L = 2_200_000, 512
    logits = torch.randn(B, 2 * L, generator=gd, device=DEV).requires_grad_(True)
    noise = torch.randn(B, L, generator=gd, device=DEV)
    assert logits.numel() > 2 ** 31
    z, _, kl, _ = F_.gauss_reparam_kl(logits, noise)
    (g,) = torch.autograd.grad(z.sum() + kl.sum(), logits)
    tail = logits[-1000:].detach().clone().requires_grad_(True)
    zt, _, klt, _ = F_.gauss_reparam_kl(tail, noise[-1000:].clone())
    (gt,) = torch.autograd.grad(zt.sum() + klt.sum(), tail)
    assert torch.equal(z[-1000:], zt) and torch.equal(kl[-1000:], klt) and torch.equal(g[-1000:], gt)
    assert bool(torch.isfinite(kl).all())
    del logits, noise, z, kl, g
    torch.cuda.empty_cache()
    # Gaussian NLL on 3x64x64 images: recon has B * 24576 = 2.4e9 elements
    Bn = 98_304
    x = torch.rand(Bn, 3, 64, 64, generator=gd, device=DEV)
    recon = torch.randn(Bn, 6, 64, 64, generator=gd, device=DEV).requires_grad_(True)
    assert recon.numel() > 2 ** 31
    kld = torch.rand(Bn, generator=gd, device=DEV)
    n, l, e, means = F_.nll_elbo(x, recon, 'gaussian', kld=kld, fuse_backward=True)
    (gr,) = torch.autograd.grad(means[0], recon)
    rt = recon[-64:].detach().clone().requires_grad_(True)
    nt, lt, et, mt = F_.nll_elbo(x[-64:].clone(), rt, 'gaussian', kld=kld[-64:].clone(), inv_count=1.0 / Bn,
                                 fuse_backward=True)
    (grt,) = torch.autograd.grad(mt[0], rt)
    assert torch.equal(n[-64:], nt) and torch.equal(l[-64:], lt)
    assert torch.equal(gr[-64:], grt)
    close(means[2], n.double().mean().float(), rtol=1e-6, atol=0)
