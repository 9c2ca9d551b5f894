"""Extreme and special values through the HIP path vs the oracle: the kernels use bare v_exp / v_log / v_rcp forms
in places, so the corners are pinned explicitly -- zero and negative Gaussian scales (legal in the
reference), saturating Bernoulli logits, non-binary targets, masked (-inf) categories, infinities and NaN
propagation.  Same tolerances as elsewhere (rtol 1e-5); inf / nan must match in position."""
import math

import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def same(a, b, rtol=1e-5, atol=1e-6):
    a, b = a.detach().float().cpu(), b.detach().float()
    assert torch.equal(torch.isnan(a), torch.isnan(b)), 'NaN positions differ'
    assert torch.equal(torch.isinf(a), torch.isinf(b)), 'inf positions differ'
    fin = torch.isfinite(b)
    assert torch.equal(torch.sign(a[~fin & ~torch.isnan(b)]), torch.sign(b[~fin & ~torch.isnan(b)]))
    torch.testing.assert_close(a[fin], b[fin], rtol=rtol, atol=atol)


def test_gaussian_special_scales():
    from vae_amd import functional as F_
    L = 8
    mu = torch.tensor([0.0, 1.0, -3.0, 1e4, -1e4, 1e-20, 5.0, 0.5])
    rows = []
    # raw s; logvar = s + 1e-6.  (|s| > 1.8e19, where s^2 overflows fp32, is left out: the reference forms
    # s^2 - log(s^2) = inf - inf = NaN there, the kernel's log(s^2) = 2 log|s| stays finite and gives +inf.)
    for s in (0.0 - 1e-6, -1.0, 1e-3, 1e-12, 1e6, 1.0e18, float('inf'), float('nan')):
        rows.append(torch.cat([mu, torch.full((L,), s)]))
    logits = torch.stack(rows)
    noise = torch.randn(len(rows), L)
    z0, p0 = O.gauss_forward(logits, noise)
    k0 = O.gauss_kl(p0)
    z1, lv1, k1, _ = F_.gauss_reparam_kl(logits.to(DEV), noise.to(DEV))
    same(z1, z0, rtol=0, atol=0)
    same(lv1, p0['gaussian']['logvar'], rtol=0, atol=0)
    same(k1, k0, rtol=1e-5, atol=1e-4)
    assert math.isinf(float(k0[0])) and float(k0[0]) > 0          # s + eps == 0: -log(0)


def test_bernoulli_nll_saturation_and_soft_targets():
    from vae_amd import functional as F_
    D = 16
    logit_vals = torch.tensor([-1e4, -200.0, -88.0, -30.0, -1e-8, 0.0, 1e-8, 30.0, 88.0, 200.0, 1e4, float('inf'),
                               float('-inf'), 17.3, -17.3, 3.0])
    xs = [torch.zeros(D), torch.ones(D), torch.full((D,), 0.3), torch.rand(D)]
    x = torch.stack(xs).view(4, 1, 4, 4)
    recon = logit_vals.expand(4, D).contiguous().view(4, 1, 4, 4)
    rc = recon.clone().requires_grad_(True)
    n0 = O.nll(x, rc, 'bernoulli')
    rd = recon.to(DEV).requires_grad_(True)
    n1 = F_.nll(x.to(DEV), rd, 'bernoulli')
    same(n1, n0, rtol=1e-5, atol=1e-3)
    # finite part only (the +-inf logits make the row infinite): per-element values through a 1-pixel "image"
    fin = torch.isfinite(logit_vals)
    for xv in (0.0, 1.0, 0.3):
        xe = torch.full((int(fin.sum()), 1, 1, 1), xv)
        re = logit_vals[fin].view(-1, 1, 1, 1)
        rc = re.clone().requires_grad_(True)
        e0 = O.nll(xe, rc, 'bernoulli')
        (g0,) = torch.autograd.grad(e0.sum(), rc)
        for fuse in (True, False):
            rd = re.to(DEV).requires_grad_(True)
            _, loss, _, means = F_.nll_elbo(xe.to(DEV), rd, 'bernoulli', fuse_backward=fuse)
            same(loss, e0, rtol=1e-5, atol=1e-6)
            (g1,) = torch.autograd.grad(means[0] * re.shape[0], rd)
            same(g1, g0, rtol=1e-5, atol=1e-7)


def test_gumbel_masked_categories_and_large_logits():
    from vae_amd import functional as F_
    torch.manual_seed(0)
    B, K = 6, 12
    logits = torch.randn(B, K)
    logits[0, 3:] = float('-inf')          # masked categories
    logits[1, :] = 1e4 * torch.randn(K)    # huge dynamic range
    logits[2, 5] = float('inf')            # one +inf: softmax is NaN there in torch as well
    logits[3, :] = -1e4
    logits[4, :] = float('-inf')           # everything masked: NaN row
    U = torch.rand(B, K)
    y0, p0 = O.gumbel_forward(logits, U, tau=0.5, training=True)
    y1, zh1, lq1 = F_.gumbel_softmax(logits.to(DEV), U.to(DEV), tau=0.5)
    same(y1, y0, rtol=1e-5, atol=1e-9)
    same(lq1, p0['discrete']['log_q_z'], rtol=1e-5, atol=1e-5)
    ok = torch.isfinite(y0).all(-1)
    assert torch.equal((zh1.cpu() == 1)[ok], (p0['discrete']['z_hard'] == 1)[ok])


def test_variance_nll_extreme_parameters():
    from vae_amd import functional as F_
    for nll_type in ('gaussian', 'disc_logistic'):
        D = 8
        x = (torch.floor(torch.rand(5, 1, 2, 4) * 256) / 255.0).clamp(0, 1)
        mean = torch.tensor([-50.0, -5.0, 0.0, 5.0, 50.0, 0.3, -0.3, 1e-6]).view(1, 1, 2, 4).expand(5, 1, 2, 4)
        var = torch.tensor([-100.0, -20.0, -4.5, 0.0, 25.0]).view(5, 1, 1, 1).expand(5, 1, 2, 4)
        recon = torch.cat([mean, var], 1).contiguous()
        n0 = O.nll(x, recon, nll_type)
        n64 = O.nll(x.double(), recon.double(), nll_type)
        n1 = F_.nll(x.to(DEV), recon.to(DEV), nll_type).cpu()
        assert torch.equal(torch.isfinite(n1), torch.isfinite(n0))
        fin = torch.isfinite(n0)
        # as close to the fp64 value as the fp32 oracle is (x2), or within rtol 1e-5 of it
        err_k, err_o = (n1.double() - n64)[fin].abs(), (n0.double() - n64)[fin].abs()
        assert bool((err_k <= 2 * err_o + 1e-5 * n64[fin].abs() + 1e-5).all()), (nll_type, n1, n0, n64)
