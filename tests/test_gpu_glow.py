"""GPU parity of the glow-side flow layers (vae_amd/flows.py: ActNorm on the BatchNormFlow kernels, InvertibleMM,
LUInvertibleMM, Shuffle, the glow / maf_split_glow builders) against fixtures produced by the reference's own
flows/ layers (tests/golden/glow.pt, flows/__init__.py:274-389, 612-625, 651-673).  rtol 1e-5 fp32 on outputs,
1e-4 on gradients (exp / slogdet forms differ from ATen's CPU ones by ulps)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def test_actnorm_golden(golden):
    from vae_amd import flows as FL
    g = golden('glow')['glow_actnorm']
    Fd = g['x'].shape[1]
    m = FL.ActNorm(Fd).to(DEV)
    y, ld = m(g['x_first'].to(DEV))                       # the first batch sets weight and bias
    assert m.initialized and ld.shape == g['logdet_first'].shape
    close(y, g['y_first'])
    close(ld, g['logdet_first'])
    for k, v in g['state_dict_after_init'].items():
        close(m.state_dict()[k], v)
    m.load_state_dict(g['state_dict_after_init'], strict=True)
    x = g['x'].to(DEV).requires_grad_(True)
    y, ld = m(x)
    close(y, g['y'])
    close(ld, g['logdet'])
    gr = torch.autograd.grad((y * g['wu'].to(DEV)).sum() + (ld * g['wl'].to(DEV)).sum(), [x, m.weight, m.bias])
    close(gr[0], g['grad_x'], rtol=1e-4)
    close(gr[1], g['grad_weight'], rtol=1e-4, atol=1e-4)
    close(gr[2], g['grad_bias'], rtol=1e-4, atol=1e-4)
    xi, ldi = m(g['y'].to(DEV), mode='inverse')
    close(xi, g['x_inverse'])
    close(ldi, g['logdet_inverse'])


@pytest.mark.parametrize('B,Fd', [(1, 1), (33, 7), (1000, 64), (257, 512), (64, 4100)])
def test_actnorm_vs_torch(B, Fd):
    """The kernel-backed map against the reference's three torch ops, parameter gradients through the column sums."""
    from vae_amd import functional as F_
    torch.manual_seed(B + Fd)
    x = (torch.randn(B, Fd) * 2 + 1).to(DEV).requires_grad_(True)
    w = (torch.randn(Fd) * 0.3).to(DEV).requires_grad_(True)
    b = torch.randn(Fd).to(DEV).requires_grad_(True)
    ld_in = torch.randn(B).to(DEV).requires_grad_(True)
    wy, wl = torch.randn(B, Fd, device=DEV), torch.randn(B, device=DEV)
    y1, l1 = F_.actnorm(x, w, b, logdet_in=ld_in)
    y0, l0 = (x - b) * torch.exp(w), ld_in + w.sum()
    close(y1, y0)
    close(l1, l0, rtol=1e-5, atol=1e-5 * Fd ** 0.5)
    g1 = torch.autograd.grad((y1 * wy).sum() + (l1 * wl).sum(), [x, w, b, ld_in])
    g0 = torch.autograd.grad((y0 * wy).sum() + (l0 * wl).sum(), [x, w, b, ld_in])
    for a, c in zip(g1, g0):
        close(a, c, rtol=1e-4, atol=1e-4 * max(1.0, float(c.abs().max())))


def test_invertible_mm_golden(golden):
    from vae_amd import flows as FL
    g = golden('glow')['glow_invmm']
    m = FL.InvertibleMM(g['x'].shape[1])
    m.load_state_dict(g['state_dict'], strict=True)
    m = m.to(DEV)
    x = g['x'].to(DEV).requires_grad_(True)
    y, ld = m(x)
    assert ld.shape == g['logdet'].shape
    close(y, g['y'])
    close(ld, g['logdet'])
    gr = torch.autograd.grad((y * g['wu'].to(DEV)).sum() + (ld * g['wl'].to(DEV)).sum(), [x, m.W])
    close(gr[0], g['grad_x'], rtol=1e-4)
    close(gr[1], g['grad_W'], rtol=1e-4, atol=1e-4)
    xi, ldi = m(g['y'].to(DEV), mode='inverse')
    close(xi, g['x_inverse'], rtol=1e-4, atol=1e-5)
    close(ldi, g['logdet_inverse'])


def test_lu_invertible_mm_golden(golden):
    from vae_amd import flows as FL
    g = golden('glow')['glow_lumm']
    m = FL.LUInvertibleMM(g['x'].shape[1])
    assert set(m.state_dict()) == set(g['state_dict'])         # P, sign_S and the masks are not checkpointed
    m.load_state_dict(g['state_dict'], strict=True)
    m.P.copy_(g['P'])
    m.sign_S.copy_(g['sign_S'])
    m = m.to(DEV)
    assert m.P.device.type == 'cuda' and m.L_mask.device.type == 'cuda'
    x = g['x'].to(DEV).requires_grad_(True)
    y, ld = m(x.to(m.L.dtype))
    close(y, g['y'])
    close(ld, g['logdet'])
    gr = torch.autograd.grad((y * g['wu'].to(DEV)).sum() + (ld * g['wl'].to(DEV)).sum(), [x, m.L, m.U, m.log_S])
    close(gr[0], g['grad_x'], rtol=1e-4)
    close(gr[1], g['grad_L'], rtol=1e-4, atol=1e-4)
    close(gr[2], g['grad_U'], rtol=1e-4, atol=1e-4)
    close(gr[3], g['grad_log_S'], rtol=1e-4, atol=1e-4)
    xi, ldi = m(g['y'].to(DEV).to(m.L.dtype), mode='inverse')
    close(xi, g['x_inverse'], rtol=1e-4, atol=1e-5)
    close(ldi, g['logdet_inverse'])


def test_shuffle_golden(golden):
    from vae_amd import flows as FL
    g = golden('glow')['glow_shuffle']
    m = FL.Shuffle(g['x'].shape[1])
    assert sorted(m.perm.tolist()) == list(range(g['x'].shape[1])) and len(m.state_dict()) == 0
    m.perm.copy_(g['perm'])
    m.inv_perm.copy_(g['inv_perm'])
    m = m.to(DEV)
    y, ld = m(g['x'].to(DEV))
    assert torch.equal(y.cpu(), g['y']) and torch.equal(ld.cpu(), g['logdet'])
    assert torch.equal(m(y, mode='inverse')[0].cpu(), g['x_inverse'])


@pytest.mark.parametrize('name', ['glow', 'mafsplitglow'])
def test_glow_stack_golden(golden, name):
    """Two-block glow / maf_split_glow stacks built by the reference: the first call (ActNorm layers initialise from
    the batch), the second call's output, accumulated logdet and every parameter gradient."""
    from vae_amd import flows as FL
    g = golden('glow')['glow_stack_' + name]
    B, Fd = g['x'].shape
    build = {'glow': lambda: FL.build_glow_flow(Fd, 16, num_blocks=2, activation_str='relu', normalization_str='none',
                                                layer_modifier='none', cuda=True),
             'mafsplitglow': lambda: FL.build_maf_split_glow_flow(Fd, 16, num_blocks=2)}[name]
    m = build()
    assert [k for k, _ in m.named_parameters()] == g['param_names']
    m.load_state_dict(g['state_dict_before'], strict=True)
    m = m.to(DEV).train(True)
    x = g['x'].to(DEV).requires_grad_(True)
    u, ld = m(x)
    assert ld.shape == g['logdet_first'].shape
    close(u, g['u_first'], rtol=1e-5, atol=2e-5)
    close(ld, g['logdet_first'], rtol=1e-5, atol=2e-5)
    sd = m.state_dict()
    for k, v in g['state_dict_after_first_call'].items():
        if 'running' not in k:
            close(sd[k], v, rtol=1e-5, atol=1e-5)
    u, ld = m(x)
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
    z = m.sample(num_samples=B, noise=u.detach())            # the inverse pass gives the inputs back
    close(z, g['x'], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('flow_type', ['glow', 'maf_split_glow'])
def test_flow_reparameterizer_glow_types(flow_type):
    """reparameterizers/flow.py:33-43: both glow flow types construct from the reference's config keys and run."""
    from vae_amd.reparameterizers import FlowReparameterizer
    cfg = {'flow_type': flow_type, 'continuous_size': 8, 'latent_size': 16, 'memory_read_steps': 1, 'cuda': True,
           'encoder_activation': 'relu', 'dense_normalization': 'none', 'encoder_layer_modifier': 'none'}
    m = FlowReparameterizer(cfg).to(DEV).train()
    x = torch.randn(64, 8, device=DEV, requires_grad=True)
    f, p = m(x)
    kl = m.kl(p)
    assert f.shape == (64, 8) and p['logdet'].shape == (64, 1) and kl.shape == (64,)
    (gx,) = torch.autograd.grad(kl.sum() + f.sum(), x)
    assert bool(torch.isfinite(gx).all())
    assert m.prior(5).shape == (5, 8)
