"""GPU parity: fused NLL + ELBO kernels vs the oracle (NLL definitions: parity unpinned by the
reference, helpers is un-vendored; the ELBO combine is pinned by tests/golden/elbo.pt)."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def close_cond(a, ref32, ref64, rtol=1e-5, atol=1e-6):
    """fp32 parity for a function that is ill-conditioned in fp32 (the discretized logistic subtracts
    two sigmoids and divides by the difference): elementwise rtol is meaningless where the fp32
    reference itself is far from the true value, so the kernel is required to be (about) as accurate
    against an fp64 evaluation as the fp32 oracle is: L2 and max error within 2x the oracle's, plus
    rtol/atol of the fp64 value."""
    a = a.detach().double().cpu()
    ref32, ref64 = ref32.detach().double(), ref64.detach().double()
    e_k, e_o = (a - ref64), (ref32 - ref64)
    assert float(e_k.norm()) <= 2 * float(e_o.norm()) + rtol * float(ref64.norm()) + atol, \
        'L2 error %g vs oracle %g' % (float(e_k.norm()), float(e_o.norm()))
    assert float(e_k.abs().max()) <= 2 * float(e_o.abs().max()) + rtol * float(ref64.abs().max()) + atol, \
        'max error %g vs oracle %g' % (float(e_k.abs().max()), float(e_o.abs().max()))


def images(B, C, H, W, nll_type):
    if nll_type == 'bernoulli':
        x = (torch.rand(B, C, H, W) < 0.13).float()
        recon = torch.randn(B, C, H, W) * 2
    else:
        x = torch.floor(torch.rand(B, C, H, W) * 256) / 255.0
        x = x.clamp(0, 1)
        recon = torch.cat([torch.randn(B, C, H, W), torch.randn(B, C, H, W) * 3 - 1.5], 1)
    return x, recon


@pytest.mark.parametrize('nll_type', ['bernoulli', 'gaussian', 'disc_logistic'])
@pytest.mark.parametrize('shape', [(1, 1, 1, 1), (5, 1, 28, 28), (64, 3, 5, 5), (300, 1, 28, 28), (3, 3, 64, 64),
                                   (17, 3, 64, 64)])
def test_nll_elbo_vs_oracle(nll_type, shape):
    from vae_amd import elbo as E
    torch.manual_seed(sum(shape))
    B = shape[0]
    x, recon = images(*shape, nll_type)
    kld = torch.rand(B) * 10
    extra = {'a': torch.rand(B), 'b': torch.rand(B)}
    mi = torch.rand(B)
    beta = 0.7
    w = torch.randn(B)

    rc = recon.clone().requires_grad_(True)
    kc = kld.clone().requires_grad_(True)
    n0 = O.nll(x, rc, nll_type)
    r0 = O.loss_function(n0, kc, beta, mut_info=mi, extra=extra)
    obj0 = (r0['loss'] * w).sum() + (r0['elbo'] * w).sum() * 0.5 + r0['loss_mean'] + 2 * r0['nll_mean'] \
        + 3 * r0['kld_mean'] + r0['elbo_mean']
    g0r, g0k = torch.autograd.grad(obj0, [rc, kc])
    # the same in fp64, to bound the fp32 oracle's own conditioning error
    rc64 = recon.double().requires_grad_(True)
    n64 = O.nll(x.double(), rc64, nll_type)
    r64 = O.loss_function(n64, kld.double(), beta, mut_info=mi.double(), extra={k: v.double() for k, v in extra.items()})
    obj64 = (r64['loss'] * w.double()).sum() + (r64['elbo'] * w.double()).sum() * 0.5 + r64['loss_mean'] \
        + 2 * r64['nll_mean'] + 3 * r64['kld_mean'] + r64['elbo_mean']
    (g64r,) = torch.autograd.grad(obj64, [rc64])

    rd = recon.to(DEV).requires_grad_(True)
    kd = kld.to(DEV).requires_grad_(True)
    r1 = E.loss_function(rd, x.to(DEV), kd, nll_type, kl_beta=beta, mut_info=mi.to(DEV),
                         extra_loss_terms={k: v.to(DEV) for k, v in extra.items()})
    assert set(r1) == set(r0)
    wd = w.to(DEV)
    obj1 = (r1['loss'] * wd).sum() + (r1['elbo'] * wd).sum() * 0.5 + r1['loss_mean'] + 2 * r1['nll_mean'] \
        + 3 * r1['kld_mean'] + r1['elbo_mean']
    g1r, g1k = torch.autograd.grad(obj1, [rd, kd])
    D = x[0].numel()
    for k in r0:
        if torch.is_tensor(r0[k]):
            close(r1[k], r0[k], rtol=1e-5, atol=1e-6 * D)
        else:
            assert r1[k] == r0[k]
    close(E.nll(x.to(DEV), rd, nll_type), n0, rtol=1e-5, atol=1e-6 * D)
    if nll_type == 'disc_logistic':
        close_cond(g1r, g0r, g64r)
    elif nll_type == 'gaussian':
        # d/dv = (1/s - d^2/s) sigmoid(v) cancels where |x - m| ~ s: accuracy judged against fp64 like the
        # oracle's own (close_cond), plus a loose elementwise bound
        close_cond(g1r, g0r, g64r)
        close(g1r, g0r, rtol=1e-4, atol=1e-5)
    else:
        close(g1r, g0r, rtol=1e-5, atol=2e-6)
    close(g1k, g0k, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', ['elbo_beta1_plain', 'elbo_beta0.25_extra'])
def test_combine_golden(golden, name):
    from vae_amd import elbo as E
    g = golden('elbo')[name]
    res = E.loss_function(None, None, g['kld'].to(DEV), 'bernoulli', kl_beta=g['kl_beta'],
                          extra_loss_terms={k: v.to(DEV) for k, v in g['extra'].items()},
                          nll_given=g['nll'].to(DEV))
    assert set(res) == set(g['result'])
    for k, v in g['result'].items():
        if torch.is_tensor(v):
            close(res[k], v, rtol=1e-6, atol=1e-5)
        else:
            assert res[k] == v


def test_combine_backward():
    from vae_amd import functional as F_
    torch.manual_seed(0)
    B = 1000
    n, k, a, m = [torch.rand(B).requires_grad_(True) for _ in range(4)]
    w = torch.randn(B)
    r0 = O.loss_function(n, k, 0.3, mut_info=m, extra={'a': a})
    g0 = torch.autograd.grad((r0['loss'] * w).sum() + r0['elbo_mean'] + r0['mut_info_mean'], [n, k, a, m])
    nd, kd, ad, md = [t.detach().to(DEV).requires_grad_(True) for t in (n, k, a, m)]
    loss, elbo, means = F_.elbo_combine(nd, kd, ad, md, kl_beta=0.3)
    g1 = torch.autograd.grad((loss * w.to(DEV)).sum() + means[1] + means[5], [nd, kd, ad, md])
    for x, y in zip(g1, g0):
        close(x, y)


def test_mixin_K_samples():
    """abstract_vae.py:347-354: K > 1 averages K decodes' NLL in training."""
    from vae_amd.elbo import ELBOMixin
    from vae_amd.reparameterizers import IsotropicGaussian
    torch.manual_seed(4)
    B, L = 16, 8

    class Model(torch.nn.Module, ELBOMixin):
        def __init__(self):
            super().__init__()
            self.config = {'continuous_size': 2 * L, 'nll_type': 'bernoulli', 'kl_beta': 1.0, 'cuda': True,
                           'half': False}
            self.reparameterizer = IsotropicGaussian(self.config)
            self.dec = torch.nn.Linear(L, 784)

        def decode(self, z):
            return self.dec(z).view(-1, 1, 28, 28)

    m = Model().to(DEV)
    m.train()
    logits = torch.randn(B, 2 * L, device=DEV)
    x = (torch.rand(B, 1, 28, 28, device=DEV) < 0.2).float()
    z, params = m.reparameterize(logits)
    out = m.loss_function(m.decode(z), x, params, K=3)
    assert out['loss'].shape == (B,) and torch.isfinite(out['loss_mean'])
    out['loss_mean'].backward()
    assert m.dec.weight.grad is not None
    # K = 1 equals the plain fused path against the oracle
    m.eval()
    z, params = m.reparameterize(logits)
    recon = m.decode(z)
    out = m.loss_function(recon, x, params)
    ref = O.loss_function(O.nll(x.cpu(), recon.detach().cpu(), 'bernoulli'),
                          O.gauss_kl(O.gauss_forward(logits.cpu(), None, training=False)[1]))
    close(out['loss'], ref['loss'], rtol=1e-5, atol=1e-3)
    close(out['elbo_mean'], ref['elbo_mean'], rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('half', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('nll_type', ['bernoulli', 'gaussian', 'disc_logistic'])
def test_bf16_nll(half, nll_type):
    torch.manual_seed(5)
    x, recon = images(32, 1, 28, 28, nll_type)
    from vae_amd import elbo as E
    n1 = E.nll(x.to(DEV).to(half), recon.to(DEV).to(half), nll_type)
    n0 = O.nll(x.to(half).float(), recon.to(half).float(), nll_type)
    assert n1.dtype == half
    if half == torch.float16:            # per-sample sums beyond fp16's range come out as inf
        big = n0 > 6e4
        assert bool(torch.isinf(n1.cpu()[big]).all())
        n1, n0 = n1.cpu().float().masked_fill(big, 0), n0.masked_fill(big, 0)
    close(n1, n0, rtol=1e-2, atol=1.0)


# ---- forward + gradient in one pass (vae_nll_elbo_fused): same results as the two-pass form ----
@pytest.mark.parametrize('nll_type', ['bernoulli', 'gaussian', 'disc_logistic'])
@pytest.mark.parametrize('shape', [(1, 1, 1, 1), (5, 1, 28, 28), (300, 1, 28, 28), (64, 3, 5, 5), (9, 3, 64, 64)])
@pytest.mark.parametrize('upstream', ['unit', 'scaled', 'mixed_means'])
def test_fused_backward_matches_two_pass_and_oracle(nll_type, shape, upstream):
    from vae_amd import elbo as E
    torch.manual_seed(sum(shape) + 1)
    B = shape[0]
    x, recon = images(*shape, nll_type)
    kld = torch.rand(B) * 10
    mi = torch.rand(B)
    extra = {'a': torch.rand(B)}
    beta = 0.3

    def objective(r):
        if upstream == 'unit':
            return r['loss_mean']                       # loss['loss_mean'].backward(): the training loop's case
        if upstream == 'scaled':
            return r['loss_mean'] * 4096.0              # a loss scaler
        return r['loss_mean'] * 0.5 + r['elbo_mean'] * 2.0 - r['nll_mean'] + r['kld_mean'] * 3.0

    rc, kc = recon.clone().requires_grad_(True), kld.clone().requires_grad_(True)
    r0 = O.loss_function(O.nll(x, rc, nll_type), kc, beta, mut_info=mi, extra=extra)
    g0 = torch.autograd.grad(objective(r0), [rc, kc])

    outs = {}
    for fuse in (True, False):
        rd, kd = recon.to(DEV).requires_grad_(True), kld.to(DEV).requires_grad_(True)
        r = E.loss_function(rd, x.to(DEV), kd, nll_type, kl_beta=beta, mut_info=mi.to(DEV),
                            extra_loss_terms={k: v.to(DEV) for k, v in extra.items()}, fuse_backward=fuse)
        g = torch.autograd.grad(objective(r), [rd, kd], retain_graph=True)
        outs[fuse] = (r, g, rd, kd)
    rf, gf, rd, kd = outs[True]
    ru, gu, _, _ = outs[False]
    r64 = None
    if nll_type != 'bernoulli':
        rc64 = recon.double().requires_grad_(True)
        r64 = O.loss_function(O.nll(x.double(), rc64, nll_type), kld.double(), beta, mut_info=mi.double(),
                              extra={k: v.double() for k, v in extra.items()})
    for k in ('loss', 'elbo', 'loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'additional_loss_mean', 'mut_info_mean'):
        close(rf[k], ru[k], rtol=1e-6, atol=1e-6)
        if nll_type == 'disc_logistic':   # log(cdf difference + 1e-7): the fp32 oracle is itself off where the difference ~ 0
            close_cond(rf[k].reshape(-1), r0[k].reshape(-1), r64[k].reshape(-1))
        else:
            close(rf[k], r0[k], rtol=1e-5, atol=1e-3 if k in ('loss', 'elbo') else 1e-4)
    close(gf[1], gu[1], rtol=1e-6, atol=0)
    close(gf[1], g0[1], rtol=1e-5, atol=1e-8)
    if nll_type == 'bernoulli':
        # sigmoid(l) - x cancels at x = 1: the fp32 reference itself is only good to an ulp of 1 times the scale
        tol = 2e-7 * float(g0[0].abs().max())
        close(gf[0], gu[0], rtol=1e-5, atol=tol)
        close(gf[0], g0[0], rtol=1e-5, atol=tol)
    else:   # ill-conditioned in fp32 (see close_cond): as accurate against fp64 as the fp32 oracle
        (g64,) = torch.autograd.grad(objective(r64), rc64)
        close_cond(gf[0], g0[0], g64)
    # a second backward through the same graph (the stored gradient is gone): general kernel, same answer
    g2 = torch.autograd.grad(objective(rf), [rd, kd])
    close(g2[0], gu[0], rtol=1e-6, atol=1e-12)
    # per-sample upstream gradients take the general kernel too
    r = E.loss_function(rd, x.to(DEV), kd, nll_type, kl_beta=beta, fuse_backward=True)
    w = torch.randn(B)
    (gw,) = torch.autograd.grad((r['loss'] * w.to(DEV)).sum() + r['loss_mean'], rd)
    rw = O.loss_function(O.nll(x, rc, nll_type), kc, beta)
    (gw0,) = torch.autograd.grad((rw['loss'] * w).sum() + rw['loss_mean'], rc)
    if nll_type == 'bernoulli':
        close(gw, gw0, rtol=1e-5, atol=2e-7 * float(gw0.abs().max()))


@pytest.mark.parametrize('B,D', [(1, 1), (7, 3), (300, 784), (4096, 100), (3, 4099)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('gm', [(1.0, 0, 0, 0, 0, 0), (0.5, 0.25, 0.25, 0, 0, 0), (2.0, 1.0, -0.5, 3.0, 0.7, -1.5)])
def test_fused_backward_one_launch_equals_finish_plus_combine(B, D, dtype, gm):
    """vae_nll_fused_backward == vae_nll_fused_grad_finish + vae_elbo_combine_bwd (the rescaled gradient bit for
    bit; the per-row constants to an ulp: the two kernels contract beta * g + g' differently)."""
    from vae_amd import _lib
    from vae_amd._lib import call, ptr
    code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
    st = _lib.stream_ptr(torch.device(DEV))
    torch.manual_seed(B + D)
    g = torch.randn(B, D).to(dtype).to(DEV)
    gmeans = torch.tensor(gm, dtype=torch.float32, device=DEV)
    a, b = g.clone(), g.clone()
    outs_a = [torch.full((B,), 7.0, dtype=dtype, device=DEV) for _ in range(3)]
    outs_b = [torch.full((B,), 9.0, dtype=dtype, device=DEV) for _ in range(3)]
    call('vae_nll_fused_grad_finish', ptr(a), ptr(gmeans), B * D, code, st)
    call('vae_elbo_combine_bwd', None, None, ptr(gmeans), 0.3, 1.0 / B, None, ptr(outs_a[0]), ptr(outs_a[1]),
         ptr(outs_a[2]), B, code, st)
    call('vae_nll_fused_backward', ptr(b), B * D, ptr(gmeans), 0.3, 1.0 / B, ptr(outs_b[0]), ptr(outs_b[1]),
         ptr(outs_b[2]), B, code, st)
    assert torch.equal(a, b)
    if gm[0] + gm[1] + gm[2] == 1.0:
        assert torch.equal(b, g)                      # unit upstream gradient: untouched
    for u, v in zip(outs_a, outs_b):
        close(v, u, rtol=1e-6 if dtype == torch.float32 else 1e-2, atol=0)
    # NULL outputs are skipped
    keep = torch.full((B,), 5.0, dtype=dtype, device=DEV)
    call('vae_nll_fused_backward', ptr(b), B * D, ptr(gmeans), 0.3, 1.0 / B, None, ptr(keep), None, B, code, st)
    close(keep, outs_a[1], rtol=1e-6 if dtype == torch.float32 else 1e-2, atol=0)


@pytest.mark.parametrize('half', [torch.bfloat16, torch.float16])
def test_fused_backward_bf16_and_no_grad(half):
    from vae_amd import elbo as E
    torch.manual_seed(4)
    x, recon = images(64, 1, 28, 28, 'bernoulli')
    rd = recon.to(DEV).to(half).requires_grad_(True)
    r = E.loss_function(rd, x.to(DEV).to(half), None, 'bernoulli')
    (g,) = torch.autograd.grad(r['loss_mean'], rd)
    rc = recon.to(half).float().requires_grad_(True)
    r0 = O.loss_function(O.nll(x, rc, 'bernoulli'), torch.zeros(64))
    (g0,) = torch.autograd.grad(r0['loss_mean'], rc)
    close(g, g0, rtol=1e-2, atol=1e-4)
    with torch.no_grad():    # no graph: the plain forward kernel, no gradient buffer
        r = E.loss_function(rd, x.to(DEV).to(half), None, 'bernoulli')
    assert not r['loss_mean'].requires_grad


def test_fused_gradient_buffer_lifetime():
    """The one-pass loss keeps a [B, D] gradient buffer from forward to backward.  It must not exist without a graph
    (no_grad / inference_mode: validation loops), must go when the outputs are dropped without a backward, and is
    handed out -- not copied -- by the backward."""
    from vae_amd import elbo as E
    torch.manual_seed(8)
    B, D = 4096, 3 * 32 * 32
    x = torch.rand(B, 3, 32, 32, device=DEV)
    recon = torch.randn(B, 3, 32, 32, device=DEV, requires_grad=True)
    nbytes = B * D * 4
    E.loss_function(recon, x, None, 'bernoulli')     # the workspace and the scalar cells exist from here on
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for ctxmgr in (torch.no_grad, torch.inference_mode):
        with ctxmgr():
            out = E.loss_function(recon, x, None, 'bernoulli')
        assert torch.cuda.memory_allocated() - base < nbytes // 2, ctxmgr
        del out
    out = E.loss_function(recon, x, None, 'bernoulli')
    assert torch.cuda.memory_allocated() - base >= nbytes          # the stored d loss_mean / d recon
    out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}   # what a logging loop keeps
    assert torch.cuda.memory_allocated() - base < nbytes // 2       # graph gone -> buffer gone
    out = E.loss_function(recon, x, None, 'bernoulli')
    (g,) = torch.autograd.grad(out['loss_mean'], recon)
    del out
    assert torch.cuda.memory_allocated() - base < nbytes + nbytes // 2   # g IS the buffer: no second [B, D] tensor


def test_autoencoder_and_nokl_losses():
    """autoencoder.py:12-32, :92-115: NLL-only objectives (dict schema of the reference, values vs the oracle)."""
    from vae_amd.elbo import AutoencoderMixin, VAENoKLMixin
    torch.manual_seed(6)
    x, recon = images(33, 1, 28, 28, 'bernoulli')
    n0 = O.nll(x, recon, 'bernoulli')

    class AE(AutoencoderMixin):
        config = {'nll_type': 'bernoulli', 'half': False}
        training = True

    class NoKL(VAENoKLMixin):
        config = {'nll_type': 'bernoulli', 'half': False}
        training = True

    rd = recon.to(DEV).requires_grad_(True)
    out = AE().loss_function(rd, x.to(DEV))
    assert set(out) == {'loss', 'elbo', 'loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'kl_beta_scalar',
                        'proxy_mean', 'mut_info_mean'}
    close(out['loss'], n0, rtol=1e-5, atol=1e-3)
    close(out['loss_mean'], n0.mean(), rtol=1e-5, atol=1e-4)
    assert AE().reparameterize(rd)[0] is rd and float(out['elbo'].abs().max()) == 0.0
    (g,) = torch.autograd.grad(out['loss_mean'], rd)
    rc = recon.clone().requires_grad_(True)
    (g0,) = torch.autograd.grad(O.nll(x, rc, 'bernoulli').mean(), rc)
    close(g, g0, rtol=1e-5, atol=2e-7 * float(g0.abs().max()))
    out = NoKL().loss_function(rd, x.to(DEV), {})
    assert set(out) == {'loss', 'loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'proxy_mean', 'mut_info_mean'}
    close(out['nll_mean'], n0.mean(), rtol=1e-5, atol=1e-4)
    assert float(out['kld_mean']) == 0.0


@pytest.mark.parametrize('dtype,rtol', [(torch.float32, 1e-5), (torch.bfloat16, 1e-2), (torch.float16, 2e-3)])
def test_elbo_combine_backward_through_functional_every_dtype(dtype, rtol):
    """functional.elbo_combine forward + backward (the K > 8 multi-sample path and loss_function(nll_given=...))
    in each element type vs the oracle: round 1 handed fp16 gradients to the kernel as bf16."""
    from vae_amd import functional as F_
    torch.manual_seed(5)
    B = 1000
    nll, kld, add, mi = torch.rand(B) * 50, torch.rand(B) * 5, torch.rand(B), torch.rand(B)
    w = torch.randn(B)
    ins = [t.to(dtype).float().requires_grad_(True) for t in (nll, kld, add, mi)]      # the values the kernel sees
    r0 = O.loss_function(ins[0], ins[1], 0.7, mut_info=ins[3], extra={'a': ins[2]})
    obj0 = (r0['loss'] * w).sum() + 0.5 * (r0['elbo'] * w).sum() + 3 * r0['loss_mean'] + r0['kld_mean']
    g0 = torch.autograd.grad(obj0, ins)
    dins = [t.to(dtype).to(DEV).requires_grad_(True) for t in (nll, kld, add, mi)]
    loss, elbo, means = F_.elbo_combine(dins[0], dins[1], dins[2], dins[3], kl_beta=0.7)
    close(loss, r0['loss'], rtol=rtol, atol=rtol)
    close(elbo, r0['elbo'], rtol=rtol, atol=rtol)
    close(means[0], r0['loss_mean'], rtol=rtol)
    wd = w.to(dtype).to(DEV)
    obj = (loss.float() * wd.float()).sum() + 0.5 * (elbo.float() * wd.float()).sum() + 3 * means[0] + means[3]
    g = torch.autograd.grad(obj, dins)
    for a, b in zip(g, g0):
        assert a.dtype == dtype
        close(a, b, rtol=rtol, atol=rtol * float(b.abs().max()))


@pytest.mark.parametrize('dtype,rtol', [(torch.bfloat16, 1e-2), (torch.float16, 2e-3), (torch.float32, 1e-5)])
def test_loss_backward_with_scaled_upstream_gradient_large_batch(dtype, rtol):
    """loss_function(fuse_backward=True).loss_mean backpropagated with an upstream gradient other than 1 (fp16 loss
    scaling) at a batch where 1/B is an fp16 subnormal: d recon must equal the general two-pass backward's (ADVICE
    r01: the one-pass kernel's stored gradient is rounded before the scale is applied; fp16 therefore keeps the
    two-pass form, bf16 / fp32 rescale)."""
    from vae_amd import elbo as E
    torch.manual_seed(11)
    B, D, scale = 65536, 48, 1024.0
    x = (torch.rand(B, 1, 6, 8) < 0.3).to(dtype).to(DEV)
    recon = (torch.randn(B, 1, 6, 8) * 2).to(dtype).to(DEV)
    kld = torch.rand(B).to(dtype).to(DEV)
    grads = []
    for fuse in (True, False):
        rd = recon.clone().requires_grad_(True)
        out = E.loss_function(rd, x, kld, 'bernoulli', fuse_backward=fuse)
        (g,) = torch.autograd.grad(out['loss_mean'] * scale, rd)
        assert g.dtype == dtype
        grads.append(g.float())
    ref = (torch.sigmoid(recon.float()) - x.float()) * (scale / B)
    for g in grads:
        close(g, ref, rtol=rtol, atol=rtol * float(ref.abs().max()) * 0.1)   # sigmoid(l) - x cancels near x = 1
    if dtype == torch.float16:
        assert torch.equal(grads[0], grads[1])     # same kernel both ways
