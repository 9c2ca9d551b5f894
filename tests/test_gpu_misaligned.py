"""Pointers that are element-aligned but not 16-byte aligned (a view one element into a buffer): every entry
point falls back to its scalar loads/stores and must give the results of the vectorised launch -- elementwise
outputs bit for bit, row sums to summation-order rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def mis(t):
    """Same values, same shape, contiguous, storage shifted by one element."""
    buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
    v = buf[1:].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 != 0 and v.is_contiguous()
    return v


def run(fn, inputs, grad_idx, exact_idx=(), rtol=2e-6):
    res = []
    for shift in (False, True):
        ins = [(mis(t) if shift else t.clone()) for t in inputs]
        ins = [t.requires_grad_(True) if i in grad_idx else t for i, t in enumerate(ins)]
        outs = [o for o in fn(*ins) if o is not None]
        w = [torch.linspace(0.5, 1.5, o.numel(), device=DEV).view(o.shape) for o in outs]
        total = sum((o.float() * wi).sum() for o, wi in zip(outs, w))
        grads = torch.autograd.grad(total, [ins[i] for i in grad_idx]) if grad_idx else []
        res.append((outs, grads))
    (o0, g0), (o1, g1) = res
    for k, (a, b) in enumerate(zip(o0, o1)):
        if k in exact_idx:
            assert torch.equal(a, b), k
        else:
            torch.testing.assert_close(b.float(), a.float(), rtol=rtol, atol=rtol * max(1.0, float(a.detach().float().abs().max())))
    for a, b in zip(g0, g1):
        torch.testing.assert_close(b.float(), a.float(), rtol=rtol, atol=rtol * max(1.0, float(a.detach().float().abs().max())))


def rnd(*shape, seed=0, rand=False, dtype=torch.float32):
    g = torch.Generator(device=DEV).manual_seed(seed)
    t = torch.rand(*shape, generator=g, device=DEV) if rand else torch.randn(*shape, generator=g, device=DEV)
    return t.to(dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,L', [(37, 64), (300, 512)])
def test_gauss_family(B, L, dtype):
    from vae_amd import functional as F_
    logits, noise = rnd(B, 2 * L, seed=1, dtype=dtype), rnd(B, L, seed=2, dtype=dtype)
    if dtype == torch.bfloat16:   # one element in = 2 bytes off: sums round to bf16, elementwise outputs stay exact
        run(lambda l, n: F_.gauss_reparam_kl(l, n)[:2], [logits, noise], [0], exact_idx=(0, 1), rtol=1e-2)
        return
    run(lambda l, n: F_.gauss_reparam_kl(l, n), [logits, noise], [0], exact_idx=(0, 1))
    run(lambda l, n: F_.gauss_reparam_multi(l, 2, noise=torch.stack([n, n * 0.5]))[:3], [logits, noise], [0],
        exact_idx=(0, 1))
    pri, npr = rnd(B, 2 * L, seed=3), rnd(B, L, seed=4)
    run(lambda e, p, a, b: F_.gauss_pair(e, p, a, b, with_prior_kl=True)[:7], [logits, pri, noise, npr], [0, 1],
        exact_idx=(0, 2, 3))
    sq, sp = rnd(B, L, seed=5, rand=True) + 0.5, rnd(B, L, seed=6, rand=True) + 0.5
    run(lambda a, b, c, d: (F_.gauss_kl(a, b, c, d),), [noise, sq, npr, sp], [0, 1, 2, 3])


@pytest.mark.parametrize('B,K', [(41, 32), (130, 1024)])
def test_discrete_family(B, K):
    from vae_amd import functional as F_
    logits, U = rnd(B, K, seed=7), rnd(B, K, seed=8, rand=True)
    run(lambda l, u: F_.gumbel_softmax(l, u, tau=0.7), [logits, U], [0], exact_idx=(1,))
    run(lambda q, p: (F_.categorical_kl(q, p),), [logits, rnd(B, K, seed=9)], [0, 1])
    run(lambda l, u: F_.relaxed_bernoulli(l, u, tau=0.6), [logits, U], [0], exact_idx=(0, 1))
    run(lambda q, p: (F_.bernoulli_kl(q, p),), [logits, rnd(B, K, seed=10)], [0, 1])
    run(lambda q: (F_.bernoulli_kl(q),), [logits], [0])


@pytest.mark.parametrize('B,L', [(29, 16), (200, 256)])
def test_beta_and_flow_kl(B, L):
    from vae_amd import functional as F_
    run(lambda l: F_.beta_split_kl(l)[:3], [rnd(B, 2 * L, seed=11)], [0], exact_idx=(0, 1))
    a1, a0 = rnd(B, L, seed=12, rand=True) + 0.3, rnd(B, L, seed=13, rand=True) + 0.3
    b1, b0 = rnd(B, L, seed=14, rand=True) + 0.5, rnd(B, L, seed=15, rand=True) + 0.5
    run(lambda a, b, c, d: (F_.beta_kl(a, b, c, d),), [a1, a0, b1, b0], [0, 1, 2, 3])
    run(lambda u, ld: (F_.flow_kl(u, ld),), [rnd(B, 2 * L, seed=16), rnd(B, 1, seed=17)], [0, 1])


@pytest.mark.parametrize('nll_type,C', [('bernoulli', 1), ('gaussian', 2), ('disc_logistic', 2)])
@pytest.mark.parametrize('B,D', [(33, 784), (5, 12288)])
def test_nll_family(nll_type, C, B, D):
    from vae_amd import functional as F_
    x = rnd(B, D, seed=18, rand=True)
    if nll_type == 'bernoulli':
        x = (x < 0.2).float()
    recon, kld = rnd(B, C * D, seed=19), rnd(B, seed=20, rand=True)
    for fuse in (False, True):
        run(lambda x_, r_, k_: F_.nll_elbo(x_, r_, nll_type, kld=k_, kl_beta=0.7, fuse_backward=fuse)[:3] +
            (F_.nll_elbo(x_, r_, nll_type, kld=k_, kl_beta=0.7, fuse_backward=fuse)[3],), [x, recon, kld], [1, 2])
    r2 = rnd(B, C * D, seed=21)
    run(lambda x_, a, b: F_.nll_multi_elbo(x_, [a, b], nll_type, kld=kld)[:3], [x, recon, r2], [1, 2])


@pytest.mark.parametrize('B,F', [(45, 64), (260, 512)])
def test_flow_tails(B, F):
    from vae_amd import functional as F_
    x, m, a = rnd(B, F, seed=22), rnd(B, F, seed=23), rnd(B, F, seed=24) * 0.3
    run(lambda x_, m_, a_: F_.made_tail(x_, m_, a_), [x, m, a], [0, 1, 2], exact_idx=(0,))
    mask = (torch.arange(F, device=DEV) % 2).float()
    run(lambda x_, s_, t_: F_.coupling_tail(x_, s_, t_, mask), [x, a, m], [0, 1, 2], exact_idx=(0,))
    lg, beta = rnd(F, seed=25) * 0.1, rnd(F, seed=26)
    run(lambda x_: F_.bnflow(x_, lg, beta, training=True, reverse=True)[:2], [x], [0])
