"""Tensors that arrive in other memory layouts than the kernels' row-major one (channels_last decoder outputs,
transposed / expanded / sliced views): the drop-in classes must give the results of the contiguous call, and
gradients must come back in a form autograd accepts for the original tensor."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


@pytest.mark.parametrize('nll_type,C', [('bernoulli', 1), ('gaussian', 2), ('disc_logistic', 2)])
@pytest.mark.parametrize('fuse', [True, False])
def test_channels_last_decoder_output(nll_type, C, fuse):
    from vae_amd.elbo import loss_function
    torch.manual_seed(0)
    B, ch, H, W = 9, 3, 8, 8
    x = torch.rand(B, ch, H, W)
    if nll_type == 'bernoulli':
        x = (x < 0.3).float()
    recon = torch.randn(B, C * ch, H, W)
    kld = torch.rand(B)
    rc = recon.clone().requires_grad_(True)
    ref = O.loss_function(O.nll(x, rc, nll_type), kld, 0.8)
    (g0,) = torch.autograd.grad(ref['loss_mean'], rc)

    rd = recon.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    assert not rd.is_contiguous()
    out = loss_function(rd, xd, kld.to(DEV), nll_type, kl_beta=0.8, fuse_backward=fuse)
    (g1,) = torch.autograd.grad(out['loss_mean'], rd)
    assert g1.shape == rd.shape
    # same bits as the row-major call on the same values
    rr = recon.to(DEV).requires_grad_(True)
    outc = loss_function(rr, x.to(DEV), kld.to(DEV), nll_type, kl_beta=0.8, fuse_backward=fuse)
    (gc,) = torch.autograd.grad(outc['loss_mean'], rr)
    assert torch.equal(out['loss'], outc['loss']) and torch.equal(g1, gc)
    if nll_type != 'disc_logistic':      # (its fp32 gradient is ill-conditioned: tests/test_gpu_elbo.py compares in fp64)
        close(out['loss'], ref['loss'], rtol=1e-5, atol=1e-4)
        close(out['loss_mean'], ref['loss_mean'], rtol=1e-5, atol=1e-4)
        close(g1, g0, rtol=1e-5, atol=1e-7)


def test_transposed_and_expanded_logits():
    from vae_amd.reparameterizers import IsotropicGaussian, GumbelSoftmax
    torch.manual_seed(1)
    B, L = 33, 16
    base = torch.randn(2 * L, B)                      # logits arrive as the transpose of a [2L, B] buffer
    noise = torch.randn(B, L)
    lc = base.t().clone().requires_grad_(True)
    z0, p0 = O.gauss_forward(lc, noise)
    k0 = O.gauss_kl(p0)
    (g0,) = torch.autograd.grad((z0 * noise).sum() + k0.sum(), lc)
    m = IsotropicGaussian({'continuous_size': 2 * L, 'cuda': True, 'half': False}).train()
    bd = base.to(DEV).requires_grad_(True)
    z1, p1 = m(bd.t(), noise=noise.to(DEV))
    k1 = m.kl(p1)
    (g1,) = torch.autograd.grad((z1 * noise.to(DEV)).sum() + k1.sum(), bd)
    assert torch.equal(z1.cpu(), z0.detach())
    close(k1, k0, atol=1e-4)
    close(g1.t(), g0, atol=1e-6)
    close(p1['gaussian']['logvar'], p0['gaussian']['logvar'], rtol=0, atol=0)

    # one logits row broadcast over the batch (expand: stride 0), as a learned prior would be
    K = 12
    row = torch.randn(1, K)
    U = torch.rand(B, K)
    rcpu = row.clone().requires_grad_(True)
    y0, q0 = O.gumbel_forward(rcpu.expand(B, K), U, tau=1.0, training=True)   # the class starts at tau0 = 1
    (gr0,) = torch.autograd.grad((y0 * U).sum(), rcpu)
    gm = GumbelSoftmax({'discrete_size': K, 'cuda': True, 'half': False}).train()
    rdev = row.to(DEV).requires_grad_(True)
    y1, q1 = gm(rdev.expand(B, K), noise=U.to(DEV))
    (gr1,) = torch.autograd.grad((y1 * U.to(DEV)).sum(), rdev)
    close(y1, y0, rtol=1e-5, atol=1e-7)
    close(gr1, gr0, rtol=1e-4, atol=1e-6)
    assert torch.equal(q1['discrete']['z_hard'].cpu() == 1, q0['discrete']['z_hard'] == 1)
