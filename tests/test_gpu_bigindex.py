"""64-bit indexing in every kernel family: launches whose tensors hold more than 2^31 elements (8.6+ GB in fp32),
forward and backward.  Each op is row-independent, so the last rows of the big launch must equal -- bit for bit --
the same rows computed in a small launch of their own.  Inputs are generated on the device; nothing is compared
with the oracle here (the CPU would need minutes): the small launches are what the oracle tests cover."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TAIL = 257


def _gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


def _check(fn, inputs, grad_inputs, row_dim=0):
    """fn(*inputs) -> tuple of row-indexed outputs.  Compare the last TAIL rows of the full launch and of its
    gradients with a launch over those rows alone."""
    full_in = [t.requires_grad_(True) if i in grad_inputs else t for i, t in enumerate(inputs)]
    outs = [o for o in fn(*full_in) if o is not None]
    total = sum(o.float().sum() for o in outs)
    grads = torch.autograd.grad(total, [full_in[i] for i in grad_inputs])
    tail_in = [t.detach()[-TAIL:].clone() for t in inputs]
    tail_in = [t.requires_grad_(True) if i in grad_inputs else t for i, t in enumerate(tail_in)]
    touts = [o for o in fn(*tail_in) if o is not None]
    ttotal = sum(o.float().sum() for o in touts)
    tgrads = torch.autograd.grad(ttotal, [tail_in[i] for i in grad_inputs])
    for o, t in zip(outs, touts):
        assert torch.equal(o.detach()[-TAIL:], t.detach())
    for g, t in zip(grads, tgrads):
        assert torch.equal(g[-TAIL:], t)
    first = outs[0].detach()
    assert bool(torch.isfinite(first[:TAIL].float()).all()) and bool(torch.isfinite(first[-TAIL:].float()).all())


@pytest.fixture(autouse=True)
def _free():
    yield
    torch.cuda.empty_cache()


B_BIG, K = 2_100_000, 1024        # 2.15e9 elements per [B, K] tensor


def test_gumbel_and_categorical_kl():
    from vae_amd import functional as F_
    g = _gen(1)
    logits = torch.randn(B_BIG, K, generator=g, device=DEV)
    assert logits.numel() > 2 ** 31
    U = torch.rand(B_BIG, K, generator=g, device=DEV)
    _check(lambda l, u: F_.gumbel_softmax(l, u, tau=0.9), [logits, U], [0])
    del U
    torch.cuda.empty_cache()
    other = torch.randn(B_BIG, K, generator=g, device=DEV)
    _check(lambda q, p: (F_.categorical_kl(q, p),), [logits.detach(), other], [0, 1])


def test_bernoulli_sample_and_kl():
    from vae_amd import functional as F_
    g = _gen(2)
    logits = torch.randn(B_BIG, K, generator=g, device=DEV)
    U = torch.rand(B_BIG, K, generator=g, device=DEV)
    _check(lambda l, u: F_.relaxed_bernoulli(l, u, tau=0.8), [logits, U], [0])
    del U
    torch.cuda.empty_cache()
    _check(lambda q: (F_.bernoulli_kl(q),), [logits.detach()], [0])


def test_beta_and_flow_kl():
    from vae_amd import functional as F_
    g = _gen(3)
    logits = torch.randn(B_BIG, K, generator=g, device=DEV)
    _check(lambda l: F_.beta_split_kl(l, 1.0)[:3], [logits], [0])
    # flow KL, 3-D form (per-row: mean over R of nlp - logdet); u [B, R, F]
    u = logits.detach().view(B_BIG, 2, K // 2)
    ld = torch.randn(B_BIG, 2, generator=g, device=DEV)
    _check(lambda a, b: (F_.flow_kl(a, b),), [u, ld], [0, 1])


def test_gauss_pair_and_multi():
    from vae_amd import functional as F_
    g = _gen(4)
    L = K // 2
    enc = torch.randn(B_BIG, K, generator=g, device=DEV)
    pri = torch.randn(B_BIG, K, generator=g, device=DEV)
    nq = torch.randn(B_BIG, L, generator=g, device=DEV)
    npr = torch.randn(B_BIG, L, generator=g, device=DEV)
    _check(lambda e, p, a, b: F_.gauss_pair(e, p, a, b, with_prior_kl=True, mean_scale=1.0)[:7], [enc, pri, nq, npr],
           [0, 1])
    del pri, nq, npr
    torch.cuda.empty_cache()
    # two draws from one read of the logits: z [S, B, L] -- rows live on dim 1
    noise = torch.randn(2, B_BIG, L, generator=g, device=DEV)
    lg = enc.detach().requires_grad_(True)
    z, lv, kl, _ = F_.gauss_reparam_multi(lg, 2, noise=noise, mean_scale=1.0)
    (gl,) = torch.autograd.grad(z.sum() + kl.sum(), lg)
    lt = enc.detach()[-TAIL:].clone().requires_grad_(True)
    zt, lvt, klt, _ = F_.gauss_reparam_multi(lt, 2, noise=noise[:, -TAIL:].clone(), mean_scale=1.0)
    (glt,) = torch.autograd.grad(zt.sum() + klt.sum(), lt)
    assert torch.equal(z[:, -TAIL:], zt) and torch.equal(kl[-TAIL:], klt) and torch.equal(gl[-TAIL:], glt)


def test_flow_tails():
    from vae_amd import functional as F_
    g = _gen(5)
    F = K
    x = torch.randn(B_BIG, F, generator=g, device=DEV)
    m = torch.randn(B_BIG, F, generator=g, device=DEV)
    a = torch.randn(B_BIG, F, generator=g, device=DEV) * 0.3
    _check(lambda x_, m_, a_: F_.made_tail(x_, m_, a_), [x, m, a], [0, 1, 2])
    mask = (torch.arange(F, device=DEV) % 2).float()
    _check(lambda x_, s_, t_: F_.coupling_tail(x_, s_, t_, mask), [x.detach(), a.detach(), m.detach()], [0, 1, 2])
    del m, a
    torch.cuda.empty_cache()
    # BatchNormFlow in eval mode (running statistics: per-row independent), Reverse folded in
    lgm = torch.randn(F, generator=g, device=DEV) * 0.1
    beta = torch.randn(F, generator=g, device=DEV)
    rm = torch.randn(F, generator=g, device=DEV) * 0.1
    rv = torch.rand(F, generator=g, device=DEV) + 0.5
    _check(lambda x_: F_.bnflow(x_, lgm, beta, rm, rv, training=False, reverse=True)[:2], [x.detach()], [0])
    # training mode: batch statistics over 2.1 M rows against a double-precision torch computation
    y, ld, bm, bv = F_.bnflow(x.detach(), lgm, beta, training=True)
    xm = x.detach().double().mean(0)
    torch.testing.assert_close(bm.double(), xm, rtol=0, atol=1e-6)
    # batch_var carries the layer's eps (flows/__init__.py:241)
    torch.testing.assert_close(bv.double(), x.detach().double().var(0, unbiased=False) + 1e-5, rtol=2e-6, atol=0)
    assert bool(torch.isfinite(y[-TAIL:]).all())
