"""GPU parity of the flow-layer tails (csrc/flowtail.hip, vae_amd/flows.py) against the oracle and against
fixtures produced by the reference's own flows/ layers (tests/golden/flowtail.pt).

Tolerance: rtol 1e-5 fp32 on every output and gradient (exp / tanh / log forms differ from ATen's by ulps),
bf16 rtol 1e-2.
"""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def load_sd(module, sd):
    module.load_state_dict({k: v for k, v in sd.items()}, strict=True)
    return module.to(DEV)


# ------------------------------------------------------------------ modules vs the reference's outputs
def test_made_golden(golden):
    from vae_amd import flows as FL
    g = golden('flowtail')['flowtail_made']
    B, Fd = g['x'].shape
    m = load_sd(FL.MADE(Fd, 16), g['state_dict'])
    x = g['x'].to(DEV).requires_grad_(True)
    u, ld = m(x)
    assert ld.shape == g['logdet'].shape
    close(u, g['u'])
    close(ld, g['logdet'])
    gr = torch.autograd.grad((u * g['wu'].to(DEV)).sum() + (ld * g['wl'].to(DEV)).sum(),
                             [x, m.trunk[3].linear.weight, m.joiner.linear.bias])
    close(gr[0], g['grad_x'])
    close(gr[1], g['grad_out_weight'])
    close(gr[2], g['grad_joiner_bias'])


@pytest.mark.parametrize('name', ['flowtail_madesplit_plain', 'flowtail_madesplit_tanh'])
def test_madesplit_golden(golden, name):
    from vae_amd import flows as FL
    g = golden('flowtail')[name]
    B, Fd = g['x'].shape
    m = load_sd(FL.MADESplit(Fd, 16, pre_exp_tanh=g['pre_exp_tanh']), g['state_dict'])
    x = g['x'].to(DEV).requires_grad_(True)
    u, ld = m(x)
    close(u, g['u'])
    close(ld, g['logdet'])
    gr = torch.autograd.grad((u * g['wu'].to(DEV)).sum() + (ld * g['wl'].to(DEV)).sum(),
                             [x, m.t_trunk[3].linear.weight, m.s_trunk[3].linear.weight])
    close(gr[0], g['grad_x'])
    close(gr[1], g['grad_t_out_weight'])
    close(gr[2], g['grad_s_out_weight'])


@pytest.mark.parametrize('name', ['flowtail_bn_train', 'flowtail_bn_eval'])
def test_bn_golden(golden, name):
    from vae_amd import flows as FL
    g = golden('flowtail')[name]
    B, Fd = g['x'].shape
    m = load_sd(FL.BatchNormFlow(Fd, momentum=g['momentum'], eps=g['eps']), g['state_dict_before'])
    m.train(g['training'])
    x = g['x'].to(DEV).requires_grad_(True)
    y, ld = m(x)
    assert ld.shape == g['logdet'].shape
    close(y, g['y'])
    close(ld, g['logdet'])
    gr = torch.autograd.grad((y * g['wu'].to(DEV)).sum() + (ld.expand(B, 1) * g['wl'].to(DEV)).sum(),
                             [x, m.log_gamma, m.beta])
    close(gr[0], g['grad_x'])
    close(gr[1], g['grad_log_gamma'])
    close(gr[2], g['grad_beta'])
    for k in ('running_mean', 'running_var'):
        close(getattr(m, k), g['state_dict_after'][k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', ['flowtail_coupling', 'flowtail_additive'])
def test_coupling_golden(golden, name):
    from vae_amd import flows as FL
    g = golden('flowtail')[name]
    B, Fd = g['x'].shape
    cls = FL.CouplingLayer if name.endswith('coupling') else FL.AdditiveCouplingLayer
    m = load_sd(cls(Fd, 16, g['mask'].to(DEV)), g['state_dict'])
    x = g['x'].to(DEV).requires_grad_(True)
    y, ld = m(x)
    close(y, g['y'])
    close(ld, g['logdet'])
    (gx,) = torch.autograd.grad((y * g['wu'].to(DEV)).sum() + (ld * g.get('wl', torch.zeros(B, 1)).to(DEV)).sum(), x,
                                retain_graph=True)
    close(gx, g['grad_x'])
    if 'grad_s_out_weight' in g:
        gs, gt = torch.autograd.grad((y * g['wu'].to(DEV)).sum() + (ld * g['wl'].to(DEV)).sum(),
                                     [m.scale_net[4].weight, m.translate_net[4].weight])
        close(gs, g['grad_s_out_weight'])
        close(gt, g['grad_t_out_weight'])


@pytest.mark.parametrize('name', ['maf', 'mafsplit', 'realnvp'])
def test_stack_golden(golden, name):
    """Two-block MAF / MAF-split / RealNVP stacks built by the reference: output, accumulated logdet, every
    parameter gradient, running statistics, log_probs and the eval-mode pass."""
    from vae_amd import flows as FL
    g = golden('flowtail')['flowtail_stack_' + name]
    B, Fd = g['x'].shape
    build = {'maf': lambda: FL.build_maf_flow(Fd, 16, num_blocks=2),
             'mafsplit': lambda: FL.build_maf_split_flow(Fd, 16, num_blocks=2),
             'realnvp': lambda: FL.build_realnvp_flow(Fd, 16, num_blocks=2, cuda=True)}[name]
    m = build()
    assert [k for k, _ in m.named_parameters()] == g['param_names']
    load_sd(m, g['state_dict_before'])
    m.train(True)
    x = g['x'].to(DEV).requires_grad_(True)
    u, ld = m(x)
    assert ld.shape == g['logdet'].shape
    close(u, g['u'], rtol=1e-5, atol=2e-5)
    close(ld, g['logdet'], rtol=1e-5, atol=2e-5)
    params = [p for p in m.parameters()]
    gr = torch.autograd.grad((u * g['wu'].to(DEV)).sum() + (ld * g['wl'].to(DEV)).sum(), [x] + params,
                             allow_unused=True)
    close(gr[0], g['grad_x'], rtol=1e-4, atol=1e-4)
    for got, want, pn in zip(gr[1:], g['grad_params'], g['param_names']):
        if want is None:
            assert got is None or float(got.abs().max()) == 0.0, pn
        else:
            close(got, want, rtol=1e-4, atol=1e-4)
    close(m.log_probs(x), g['log_probs_second_train_call'], rtol=1e-5, atol=1e-4)
    sd = m.state_dict()
    for k, v in g['state_dict_after_train'].items():
        if 'running' in k:
            close(sd[k], v, rtol=1e-5, atol=1e-6)
    m.eval()
    ue, lde = m(x)
    close(ue, g['u_eval'], rtol=1e-5, atol=2e-5)
    close(lde, g['logdet_eval'], rtol=1e-5, atol=2e-5)


# ------------------------------------------------------------------ kernels vs the oracle, many shapes
@pytest.mark.parametrize('B,Fd', [(1, 1), (5, 3), (33, 64), (257, 100), (1024, 512), (300, 1028), (64, 4096)])
@pytest.mark.parametrize('tanh', [False, True])
def test_made_tail_vs_oracle(B, Fd, tanh):
    from vae_amd import functional as F_
    torch.manual_seed(B + Fd)
    x, ma, ld0 = torch.randn(B, Fd), torch.randn(B, 2 * Fd), torch.randn(B)
    wu, wl = torch.randn(B, Fd), torch.randn(B)
    xc, mac, lc = [t.clone().requires_grad_(True) for t in (x, ma, ld0)]
    u0, l0 = O.made_tail(xc, mac[:, :Fd], mac[:, Fd:], pre_exp_tanh=tanh)
    l0 = lc + l0.squeeze(-1)
    g0 = torch.autograd.grad((u0 * wu).sum() + (l0 * wl).sum(), [xc, mac, lc])
    xd, mad, ld = [t.to(DEV).requires_grad_(True) for t in (x, ma, ld0)]
    if tanh:   # the split form: two separate tensors
        md, ad = mad[:, :Fd].contiguous(), mad[:, Fd:].contiguous()
        u1, l1 = F_.made_tail(xd, md, ad, logdet_in=ld, pre_exp_tanh=True)
    else:      # the packed form
        u1, l1 = F_.made_tail(xd, mad, None, logdet_in=ld)
    close(u1, u0)
    close(l1, l0, rtol=1e-5, atol=1e-5 * max(Fd, 1) ** 0.5)
    g1 = torch.autograd.grad((u1 * wu.to(DEV)).sum() + (l1 * wl.to(DEV)).sum(), [xd, mad, ld])
    for a, b in zip(g1, g0):
        close(a, b)


@pytest.mark.parametrize('B,Fd', [(1, 2), (7, 6), (129, 64), (512, 500), (1000, 1024)])
@pytest.mark.parametrize('additive', [False, True])
def test_coupling_tail_vs_oracle(B, Fd, additive):
    from vae_amd import functional as F_
    torch.manual_seed(B * 3 + Fd)
    x, s, t, ld0 = torch.randn(B, Fd), torch.randn(B, Fd) * 0.5, torch.randn(B, Fd), torch.randn(B)
    mask = (torch.arange(Fd) % 2).float()
    wu, wl = torch.randn(B, Fd), torch.randn(B)
    xc, sc, tc, lc = [v.clone().requires_grad_(True) for v in (x, s, t, ld0)]
    y0, l0 = O.coupling_tail(xc, None if additive else sc, tc, mask)
    l0 = lc + l0.squeeze(-1)
    ins0 = [xc, tc, lc] if additive else [xc, sc, tc, lc]
    g0 = torch.autograd.grad((y0 * wu).sum() + (l0 * wl).sum(), ins0)
    xd, sd, td, ld = [v.to(DEV).requires_grad_(True) for v in (x, s, t, ld0)]
    y1, l1 = F_.coupling_tail(xd, None if additive else sd, td, mask.to(DEV), logdet_in=ld)
    close(y1, y0)
    close(l1, l0, rtol=1e-5, atol=1e-5 * Fd ** 0.5)
    ins1 = [xd, td, ld] if additive else [xd, sd, td, ld]
    g1 = torch.autograd.grad((y1 * wu.to(DEV)).sum() + (l1 * wl.to(DEV)).sum(), ins1)
    for a, b in zip(g1, g0):
        close(a, b)


@pytest.mark.parametrize('B,Fd', [(2, 1), (8, 6), (100, 64), (1000, 100), (4096, 512), (777, 1024), (300, 1022),
                                  (513, 2048), (260, 4096), (70, 4100), (33, 9000), (5, 1030), (12, 16384 + 8)])
@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('reverse', [False, True])
def test_bnflow_vs_oracle(B, Fd, train, reverse):
    """Batch statistics with a common offset several std wide (|mean| up to ~20, std 0.5..2: the shifted sums must
    not cancel),
    normalise, logdet, the fused Reverse, and the full backward (through the batch statistics)."""
    from vae_amd import functional as F_
    torch.manual_seed(B + 7 * Fd)
    x = torch.randn(B, Fd) * (0.5 + 1.5 * torch.rand(Fd)) + 5.0 * torch.randn(Fd)
    lg, beta = torch.randn(Fd) * 0.3, torch.randn(Fd)
    rm, rv = 5.0 * torch.randn(Fd), 0.5 + torch.rand(Fd)
    ld0 = torch.randn(B)
    wu, wl = torch.randn(B, Fd), torch.randn(B)
    xc, lgc, bc, lc = [v.clone().requires_grad_(True) for v in (x, lg, beta, ld0)]
    y0, l0, mean0, var0 = O.bnflow(xc, lgc, bc, rm, rv, training=train)
    if reverse:
        y0 = O.reverse_columns(y0)
    l0 = lc + l0
    g0 = torch.autograd.grad((y0 * wu).sum() + (l0 * wl).sum(), [xc, lgc, bc, lc])
    xd, lgd, bd, ld = [v.to(DEV).requires_grad_(True) for v in (x, lg, beta, ld0)]
    y1, l1, mean1, var1 = F_.bnflow(xd, lgd, bd, rm.to(DEV), rv.to(DEV), logdet_in=ld, training=train, reverse=reverse)
    close(mean1, mean0, rtol=1e-6, atol=1e-5)
    close(var1, var0, rtol=2e-5, atol=1e-7)
    # an ulp of the fp32 batch mean (|mean| ~ 20: 2e-6) shows up in y divided by the std
    close(y1, y0, rtol=1e-5, atol=3e-5)
    close(l1, l0, rtol=1e-5, atol=1e-5 * Fd ** 0.5)
    g1 = torch.autograd.grad((y1 * wu.to(DEV)).sum() + (l1 * wl.to(DEV)).sum(), [xd, lgd, bd, ld])
    scale = float(g0[0].abs().max())
    close(g1[0], g0[0], rtol=1e-4, atol=1e-4 * max(scale, 1.0))
    close(g1[1], g0[1], rtol=1e-4, atol=1e-4 * max(float(g0[1].abs().max()), 1.0))
    close(g1[2], g0[2], rtol=1e-4, atol=1e-4 * max(float(g0[2].abs().max()), 1.0))
    close(g1[3], g0[3])


@pytest.mark.parametrize('half', [torch.bfloat16, torch.float16])
def test_flowtail_bf16(half):
    from vae_amd import functional as F_
    torch.manual_seed(5)
    B, Fd = 256, 128
    x, ma = torch.randn(B, Fd).to(half), torch.randn(B, 2 * Fd).to(half)
    u0, l0 = O.made_tail(x.float(), ma[:, :Fd].float(), ma[:, Fd:].float())
    u1, l1 = F_.made_tail(x.to(DEV), ma.to(DEV))
    assert u1.dtype == half
    close(u1, u0, rtol=1e-2, atol=1e-2)
    close(l1, l0.squeeze(-1), rtol=1e-2, atol=0.2)
    lg, beta = torch.randn(Fd) * 0.3, torch.randn(Fd)
    y0, ld0, _, _ = O.bnflow(x.float(), lg, beta, None, None, training=True)
    y1, ld1, _, _ = F_.bnflow(x.to(DEV), lg.to(DEV), beta.to(DEV), training=True, reverse=True)
    close(y1, O.reverse_columns(y0), rtol=2e-2, atol=2e-2)
    close(ld1, ld0.expand(B), rtol=1e-2, atol=0.2)
    # coupling tail + gradients of all three tails against fp32 autograd on the same 16-bit values
    s_raw, t_raw = (torch.randn(B, Fd) * 0.3).to(half), torch.randn(B, Fd).to(half)
    mask = (torch.arange(Fd) % 2).float()
    xc, sc, tc = x.float().requires_grad_(True), s_raw.float().requires_grad_(True), t_raw.float().requires_grad_(True)
    yc0, lc0 = O.coupling_tail(xc, sc, tc, mask)
    gc0 = torch.autograd.grad(yc0.sum() + lc0.sum(), [xc, sc, tc])
    xd, sd, td = [t.to(DEV).requires_grad_(True) for t in (x, s_raw, t_raw)]
    yc1, lc1 = F_.coupling_tail(xd, sd, td, mask.to(DEV))
    close(yc1, yc0, rtol=1e-2, atol=1e-2)
    gc1 = torch.autograd.grad(yc1.float().sum() + lc1.float().sum(), [xd, sd, td])
    # d x here is the direct path only (x * mask also feeds the nets through stock autograd): compare that part
    close(gc1[0], torch.exp(s_raw.float() * (1 - mask)), rtol=2e-2, atol=2e-2)
    close(gc1[1], gc0[1], rtol=2e-2, atol=2e-2 * float(gc0[1].abs().max()))
    close(gc1[2], gc0[2], rtol=2e-2, atol=2e-2)


def test_flowtail_empty_and_limits():
    from vae_amd import functional as F_
    u, ld = F_.made_tail(torch.zeros(0, 8, device=DEV), torch.zeros(0, 16, device=DEV))
    assert u.shape == (0, 8) and ld.shape == (0,)
    with pytest.raises(ValueError):   # parameter vectors of the wrong width
        F_.bnflow(torch.zeros(4, 64, device=DEV), torch.zeros(32, device=DEV), torch.zeros(64, device=DEV))


def test_flow_reparameterizer_with_kernel_backed_maf():
    """FlowReparameterizer over a vae_amd.flows MAF stack: forward + the flow 'KL' + backward run end to end
    (the reference's reparameterizers/flow.py:80-92 + :67-78 with flows.build_maf_flow)."""
    from vae_amd import flows as FL
    from vae_amd.reparameterizers import FlowReparameterizer
    torch.manual_seed(9)
    B, Fd = 512, 64
    flow = FL.build_maf_flow(Fd, 128, num_blocks=3).to(DEV)
    m = FlowReparameterizer({'continuous_size': Fd, 'flow_module': flow})
    m.train()
    x = torch.randn(B, Fd, device=DEV, requires_grad=True)
    z, params = m(x)
    kl = m.kl(params)
    assert z.shape == (B, Fd) and kl.shape == (B,) and params['logdet'].shape == (B, 1)
    # the same stack, layer by layer, in plain torch on the CPU copy of the weights
    cpu = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
    h, ld = x.detach().cpu(), torch.zeros(B, 1)
    for i in range(3):
        p = '%d.' % (3 * i)
        w = lambda n: cpu[p + n]
        a1 = torch.nn.functional.linear(h, w('joiner.linear.weight') * w('joiner.mask'), w('joiner.linear.bias'))
        a2 = torch.nn.functional.linear(torch.relu(a1), w('trunk.1.linear.weight') * w('trunk.1.mask'),
                                        w('trunk.1.linear.bias'))
        ma = torch.nn.functional.linear(torch.relu(a2), w('trunk.3.linear.weight') * w('trunk.3.mask'),
                                        w('trunk.3.linear.bias'))
        h, l1 = O.made_tail(h, ma[:, :Fd], ma[:, Fd:])
        q = '%d.' % (3 * i + 1)
        h, l2, _, _ = O.bnflow(h, cpu[q + 'log_gamma'], cpu[q + 'beta'], None, None, training=True)
        h = O.reverse_columns(h)
        ld = ld + l1 + l2
    close(z, h, rtol=1e-4, atol=1e-4)
    close(params['logdet'], ld, rtol=1e-4, atol=1e-3)
    close(kl, O.flow_kl({'recon_x': h, 'logdet': ld}), rtol=1e-4, atol=1e-3)
    (gx,) = torch.autograd.grad(kl.sum() + z.sum(), x)
    assert bool(torch.isfinite(gx).all())


@pytest.mark.parametrize('B,F', [(7, 5), (300, 64), (1025, 512)])
@pytest.mark.parametrize('reverse', [False, True])
def test_bn_split_phases_equal_the_single_call(B, F, reverse):
    """The cross-rank form of BatchNormFlow (statistics-only forward, reduction, apply; backward sums, reduction,
    apply) with an identity reduction and count = B is the one-device layer."""
    from vae_amd import functional as F_
    torch.manual_seed(B + F)
    x = (torch.randn(B, F) * 1.7 + 0.4).to(DEV)
    lg, beta = (torch.randn(F) * 0.2).to(DEV), torch.randn(F).to(DEV)
    ld_in = torch.randn(B).to(DEV)
    w, v = torch.randn(B, F).to(DEV), torch.randn(B).to(DEV)
    calls = []
    res = []
    for sync in (False, True):
        xi, lgi, bi = x.clone().requires_grad_(True), lg.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        kw = dict(reduce_fn=lambda t: calls.append(tuple(t.shape)) or t, count=B) if sync else {}
        y, ld, m, var = F_.bnflow(xi, lgi, bi, logdet_in=ld_in, training=True, reverse=reverse, **kw)
        g = torch.autograd.grad((y * w).sum() + (ld * v).sum(), [xi, lgi, bi])
        res.append((y, ld, m, var) + g)
    assert calls == [(2 * F + 1,), (2 * F + 1,)]
    for a, b in zip(*res):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-5 * max(1.0, float(a.abs().max())))


# ------------------------------------------------------------------ MADE tail that also leaves the BatchNormFlow moments
@pytest.mark.parametrize('B,Fd', [(1, 4), (2, 8), (37, 64), (300, 128), (1025, 512), (4096, 512), (777, 1024),
                                  (513, 2048), (260, 4096), (5000, 36)])
@pytest.mark.parametrize('split', [False, True])
@pytest.mark.parametrize('reverse', [False, True])
def test_made_tail_moments_then_bnflow_equals_the_two_plain_calls(B, Fd, split, reverse):
    """vae_made_tail_fwd_moments + vae_bnflow_fwd(training == 2) against the two plain launches: u and y bit for bit
    (same shift, same partial geometry, same summation order), logdet to an ulp of the row sum, the same gradients; and
    both against the oracle (O.made_tail, O.bnflow)."""
    from vae_amd import functional as F_
    torch.manual_seed(B * 5 + Fd)
    x, ma, ld0 = torch.randn(B, Fd) * 2 + 3, torch.randn(B, 2 * Fd) * 0.7, torch.randn(B)
    lg, beta = torch.randn(Fd) * 0.3, torch.randn(Fd)
    wu, wl = torch.randn(B, Fd).to(DEV), torch.randn(B).to(DEV)
    res = []
    for fold in (False, True):
        xd, mad, ldd, lgd, bd = [t.to(DEV).requires_grad_(True) for t in (x, ma, ld0, lg, beta)]
        stats = F_.bn_stats_buffer(xd) if fold else None
        assert not fold or stats is not None
        if split:
            md, ad = mad[:, :Fd].contiguous(), mad[:, Fd:].contiguous()
            u, l = F_.made_tail(xd, md, ad, logdet_in=ldd, pre_exp_tanh=True, bn_stats=stats)
        else:
            u, l = F_.made_tail(xd, mad, None, logdet_in=ldd, bn_stats=stats)
        y, l2, mean, var = F_.bnflow(u, lgd, bd, logdet_in=l, training=True, reverse=reverse, moments=stats)
        g = torch.autograd.grad((y * wu).sum() + (l2 * wl).sum(), [xd, mad, ldd, lgd, bd])
        res.append((u, y, mean, var, l, l2) + g)
    plain, folded = res
    for k in range(4):
        assert torch.equal(plain[k], folded[k]), k
    for a, b in zip(plain[4:], folded[4:]):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-5 * max(1.0, float(a.abs().max())))
    # the oracle
    u0, l0 = O.made_tail(x, ma[:, :Fd], ma[:, Fd:], pre_exp_tanh=split)
    y0, lb0, mean0, var0 = O.bnflow(u0, lg, beta, None, None, training=True)
    if reverse:
        y0 = O.reverse_columns(y0)
    close(folded[0], u0, rtol=1e-5, atol=1e-5)
    close(folded[2], mean0, rtol=1e-5, atol=1e-5)
    close(folded[3], var0, rtol=5e-5, atol=1e-6)
    if B > 1:
        close(folded[1], y0, rtol=1e-4, atol=1e-4)
    close(folded[5], ld0 + l0.squeeze(-1) + lb0, rtol=1e-5, atol=2e-5 * Fd ** 0.5)


@pytest.mark.parametrize('half', [torch.bfloat16, torch.float16])
def test_made_tail_moments_16bit(half):
    from vae_amd import functional as F_
    torch.manual_seed(11)
    B, Fd = 1500, 256
    x, ma = torch.randn(B, Fd).to(half).to(DEV), (torch.randn(B, 2 * Fd) * 0.5).to(half).to(DEV)
    lg, beta = (torch.randn(Fd) * 0.3).to(DEV), torch.randn(Fd).to(DEV)
    u0, l0 = F_.made_tail(x, ma)
    y0, ll0, m0, v0 = F_.bnflow(u0, lg, beta, logdet_in=l0, training=True, reverse=True)
    stats = F_.bn_stats_buffer(x)
    u1, l1 = F_.made_tail(x, ma, bn_stats=stats)
    y1, ll1, m1, v1 = F_.bnflow(u1, lg, beta, logdet_in=l1, training=True, reverse=True, moments=stats)
    assert torch.equal(u0, u1) and torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    close(ll1, ll0, rtol=1e-2, atol=0.5)


def test_made_tail_moments_unsupported_shapes_and_modes():
    from vae_amd import functional as F_
    from vae_amd import _lib
    lib = _lib.load()
    assert lib.vae_made_tail_moments_supported(512, 0) == 1
    assert lib.vae_made_tail_moments_supported(510, 0) == 0       # not a whole number of 16-byte vectors
    assert lib.vae_made_tail_moments_supported(4 * 4 * 256 + 4, 0) == 0   # wider than one column window
    assert F_.bn_stats_buffer(torch.zeros(4, 510, device=DEV)) is None
    # eval mode / cross-rank statistics ignore a moments buffer (the running statistics / the split phases are used)
    x = torch.randn(64, 32, device=DEV)
    lg, beta = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
    junk = torch.full((lib.vae_bnflow_stats_floats(32),), float('nan'), device=DEV)
    y, ld, _, _ = F_.bnflow(x, lg, beta, torch.zeros(32, device=DEV), torch.ones(32, device=DEV), training=False,
                            moments=junk)
    assert bool(torch.isfinite(y).all())


def test_flow_sequential_pairs_made_with_batchnorm(monkeypatch):
    """FlowSequential routes MADE -> BatchNormFlow through the folded launch in training (one launch fewer per block) and
    the result equals the unpaired layers'."""
    from vae_amd import flows, functional as F_
    torch.manual_seed(3)
    net = flows.build_maf_flow(64, 128, num_blocks=3).to(DEV).train()
    x = torch.randn(512, 64, device=DEV)
    seen = []
    real = F_.made_tail

    def spy(*a, **k):
        seen.append(k.get('bn_stats') is not None)
        return real(*a, **k)
    monkeypatch.setattr(F_, 'made_tail', spy)
    u1, ld1 = net(x)
    assert seen and all(seen)
    seen.clear()
    monkeypatch.setattr(F_, 'bn_stats_buffer', lambda x: None)
    u0, ld0 = net(x)
    assert seen and not any(seen)
    torch.testing.assert_close(u1, u0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ld1, ld0, rtol=1e-5, atol=1e-4)
    net.eval()
    seen.clear()
    net(x)
    assert not any(seen)


@pytest.mark.parametrize('half', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('Fd', [2048, 4096, 9008])
def test_bnflow_16bit_wide_rows(half, Fd):
    """16-bit BatchNormFlow on rows of 256..1126 eight-element vectors: the column moments are taken in windows of
    two vectors per lane (four 8-element vectors per lane do not fit the register file), forward and backward."""
    from vae_amd import functional as F_
    torch.manual_seed(Fd)
    B = 260
    x = (torch.randn(B, Fd) * (0.5 + torch.rand(Fd)) + torch.randn(Fd)).to(half)
    lg, beta = torch.randn(Fd) * 0.2, torch.randn(Fd)
    w = torch.randn(B, Fd)
    xc, lgc, bc = x.float().requires_grad_(True), lg.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y0, ld0, mean0, var0 = O.bnflow(xc, lgc, bc, None, None, training=True)
    g0 = torch.autograd.grad((y0 * w).sum(), [xc, lgc, bc])
    xd, lgd, bd = x.to(DEV).requires_grad_(True), lg.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    y1, ld1, mean1, var1 = F_.bnflow(xd, lgd, bd, training=True)
    g1 = torch.autograd.grad((y1.float() * w.to(DEV)).sum(), [xd, lgd, bd])
    close(mean1, mean0, rtol=1e-4, atol=1e-4)
    close(var1, var0, rtol=1e-3, atol=1e-5)
    close(y1, y0, rtol=2e-2, atol=2e-2)
    close(g1[1], g0[1], rtol=3e-2, atol=0.5)
    close(g1[2], g0[2], rtol=3e-2, atol=0.5)
    close(g1[0], g0[0], rtol=5e-2, atol=5e-2 * float(g0[0].abs().max()))
