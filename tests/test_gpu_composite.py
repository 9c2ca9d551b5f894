"""GPU parity: Concat / Sequential / Mixture composition and the VRNN per-step path."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def cfg(**kw):
    c = {'continuous_size': 8, 'discrete_size': 5, 'cuda': True, 'half': False, 'epochs': 10,
         'continuous_mut_info': 0, 'discrete_mut_info': 0, 'kl_beta': 1.0, 'use_prior_kl': False,
         'reparam_type': 'isotropic_gaussian', 'nll_type': 'bernoulli'}
    c.update(kw)
    return c


@pytest.mark.parametrize('name', ['concat_train', 'concat_eval'])
def test_concat_golden(golden, name):
    from vae_amd.reparameterizers import ConcatReparameterizer
    g = golden('concat')[name]
    m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'], cfg())
    m.train(g['training'])
    assert m.input_size == g['input_size'] and m.output_size == g['output_size']
    assert m.get_reparameterizer_scalars() == g['scalars']
    logits = g['logits'].to(DEV).requires_grad_(True)
    noise = [None if g['noise'] is None else g['noise'].to(DEV), g['U'].to(DEV)]
    z, params = m(logits, noise=noise)
    assert set(params) == {'logits', 'concat'} and len(params['concat']) == 2
    assert set(params['concat'][0]) == set(g['params']['concat'][0])
    assert set(params['concat'][1]) == set(g['params']['concat'][1])
    L = 4
    assert torch.equal(z[:, :L].detach().cpu(), g['z'][:, :L])                  # gaussian part bit-exact
    if g['training']:
        close(z[:, L:], g['z'][:, L:], rtol=1e-5, atol=1e-7)
    else:
        assert torch.equal(z[:, L:].detach().cpu() == 1, g['z'][:, L:] == 1)     # hard one-hot part
    for mine, ref in zip(params['concat'], g['params']['concat']):
        assert torch.equal(mine['logits'].detach().cpu(), ref['logits'])
    kl = m.kl(params)
    close(kl, g['kl'], rtol=1e-5, atol=1e-5)
    (gl,) = torch.autograd.grad((z * g['w'].to(DEV)).sum() + kl.sum(), logits)
    close(gl, g['grad_logits'], rtol=1e-4, atol=1e-5)


def test_concat_large_vs_oracle():
    from vae_amd.reparameterizers import ConcatReparameterizer
    torch.manual_seed(0)
    B, L, K = 2048, 64, 256
    m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'], cfg(continuous_size=2 * L, discrete_size=K))
    m.eval()
    logits = torch.randn(B, 2 * L + K)
    U = torch.rand(B, K)
    z0, p0 = O.concat_forward(logits, [('isotropic_gaussian', 2 * L, {}), ('discrete', K, {'tau': 1.0})],
                              [None, U], training=False)
    z1, p1 = m(logits.to(DEV), noise=[None, U.to(DEV)])
    assert torch.equal(z1[:, :L].cpu(), z0[:, :L])
    assert torch.equal(z1[:, L:].cpu() == 1, z0[:, L:] == 1)                      # every hard index
    close(m.kl(p1), O.concat_kl(p0, ['isotropic_gaussian', 'discrete']), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('B,L,K,Kb', [(7, 3, 5, 2), (300, 64, 128, 32), (1025, 36, 10, 20), (64, 512, 1024, 8)])
@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_concat_fused_matches_per_child_path(B, L, K, Kb, train, dtype, monkeypatch):
    """The one-node Concat (children write strided slices of shared buffers) against the per-child
    path (narrow views + torch.cat) on the same injected noise: values and d(logits) bit-equal, since
    both run the same kernels and only the addressing differs."""
    from vae_amd.reparameterizers import ConcatReparameterizer, concat_reparameterizer as C
    torch.manual_seed(B + L)
    c = cfg(continuous_size=2 * L, discrete_size=K)
    kinds = ['isotropic_gaussian', 'discrete', 'bernoulli']
    W = 2 * L + K + K
    logits = torch.randn(B, W, device=DEV).to(dtype)
    noise = [torch.randn(B, L, device=DEV).to(dtype) if train else None,
             torch.rand(B, K, device=DEV).clamp_(1e-3, 1 - 1e-3).to(dtype),
             torch.rand(B, K, device=DEV).clamp_(1e-3, 1 - 1e-3).to(dtype)]
    w = torch.randn(B, L + 2 * K, device=DEV).to(dtype)

    def run(fused):
        if not fused:
            monkeypatch.setattr(C, '_FUSABLE', ())
        m = ConcatReparameterizer(kinds, dict(c))
        m.train(train)
        x = logits.clone().requires_grad_(True)
        z, params = m(x, noise=noise)
        kl = m.kl(params)
        extra = sum((p['discrete']['z_hard'].float() * 0.25).sum() + (p['z'].float() * 0.5).sum()
                    for p in params['concat'] if 'discrete' in p)
        extra = extra + params['concat'][1]['discrete']['log_q_z'].float().sum() * 0.125 \
            + params['concat'][0]['gaussian']['logvar'].float().sum() * 0.0625 \
            + params['concat'][0]['gaussian']['mu_mean'].float() + params['concat'][0]['gaussian']['logvar_mean'].float()
        (g,) = torch.autograd.grad((z.float() * w.float()).sum() + kl.float().sum() + extra, x)
        monkeypatch.undo()
        return z, params, kl, g

    z1, p1, kl1, g1 = run(True)
    z0, p0, kl0, g0 = run(False)
    assert z1.shape == (B, L + 2 * K) and z1.is_contiguous()
    assert torch.equal(z1, z0) and torch.equal(kl1, kl0)
    for a, b in zip(p1['concat'], p0['concat']):
        assert set(a) == set(b)
        assert torch.equal(a['z'], b['z']) and torch.equal(a['logits'], b['logits'])
        if 'discrete' in a:
            assert set(a['discrete']) == set(b['discrete'])
            assert torch.equal(a['discrete']['z_hard'], b['discrete']['z_hard'])
        else:
            assert set(a['gaussian']) == set(b['gaussian'])
            assert torch.equal(a['gaussian']['logvar'], b['gaussian']['logvar'])
            assert torch.equal(a['gaussian']['mu_mean'], b['gaussian']['mu_mean'])
    # fp32 training: same kernels, same operands -> identical.  In eval the per-child path returns z = the mu
    # VIEW, so autograd adds dz to the kernel's d(mu) afterwards (another rounding order); in bf16 the
    # per-child path sums its narrow-backward contributions in bf16.
    if dtype == torch.float32 and train:
        assert torch.equal(g1, g0)
    elif dtype == torch.float32:
        close(g1, g0, rtol=1e-6, atol=1e-6)
    else:
        close(g1, g0, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('B,L,K', [(300, 64, 128), (1025, 36, 12), (64, 512, 1024), (77, 8, 256), (5000, 128, 256),
                                   (33, 1024, 32)])
@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('inject', [True, False])
def test_concat_row_kernel_matches_per_child_launches(B, L, K, train, dtype, inject, monkeypatch):
    """Concat(Gaussian, Gumbel) through the one-launch row kernel (csrc/concat.hip: a lane group owns the whole
    [2L+K] row) against the per-child launches over strided slices, same noise (injected, or the same Philox
    streams): the Gaussian part and the KL's inputs bit for bit (same roundings), the softmax part to the summation
    order of the row (another lane-group width), the hard sample's hot positions identical, d(logits) likewise."""
    from vae_amd import functional as F_
    from vae_amd import rng
    from vae_amd.reparameterizers import ConcatReparameterizer
    if dtype != torch.float32 and (L % 8 or K % 8):
        # 16-bit rows that are not whole 16-byte vectors: the row kernel must decline (the module then takes the
        # per-child launches, which test_concat_fused_matches_per_child_path covers for these shapes)
        probe = torch.empty(2, 2 * L + K, device=DEV, dtype=dtype)
        assert not F_._row_kernel_ok(probe, (('gauss', L, 1e-6, None), ('gumbel', K, 1.0)))
        return
    torch.manual_seed(B + L + K)
    c = cfg(continuous_size=2 * L, discrete_size=K)
    logits = torch.randn(B, 2 * L + K, device=DEV).to(dtype)
    noise = [torch.randn(B, L, device=DEV).to(dtype) if train else None,
             torch.rand(B, K, device=DEV).clamp_(1e-3, 1 - 1e-3).to(dtype)] if inject else None
    w = torch.randn(B, L + K, device=DEV).to(dtype)
    assert F_._row_kernel_ok(logits, (('gauss', L, 1e-6, None), ('gumbel', K, 1.0)))

    def run(row_kernel):
        monkeypatch.setattr(F_, 'ROW_KERNEL', row_kernel)
        m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'], dict(c))
        m.train(train)
        rng.manual_seed(11)
        x = logits.clone().requires_grad_(True)
        z, params = m(x, noise=noise)
        kl = m.kl(params)
        g0 = params['concat'][0]['gaussian']
        obj = (z.float() * w.float()).sum() + (kl.float() * 0.5).sum() + g0['mu_mean'].float() + 2 * g0['logvar_mean'].float()
        (g,) = torch.autograd.grad(obj, x)
        return z, params, kl, g

    z1, p1, kl1, g1 = run(True)
    z0, p0, kl0, g0 = run(False)
    tol = dict(rtol=2e-6, atol=1e-7) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-3)
    assert z1.shape == (B, L + K) and z1.is_contiguous()
    assert torch.equal(z1[:, :L], z0[:, :L])                                     # Gaussian sample: bit for bit
    if train:
        close(z1[:, L:], z0[:, L:], **tol)
    else:
        assert torch.equal(z1[:, L:] == 1, z0[:, L:] == 1)                       # every hot position
        close(p1['concat'][1]['z'], p0['concat'][1]['z'], **tol)
    close(kl1, kl0, rtol=1e-5 if dtype == torch.float32 else 1e-2, atol=1e-4 if dtype == torch.float32 else 0.3)
    d1, d0 = p1['concat'][1]['discrete'], p0['concat'][1]['discrete']
    assert set(d1) == set(d0)
    assert torch.equal(d1['z_hard'] == 1, d0['z_hard'] == 1)
    close(d1['log_q_z'], d0['log_q_z'], rtol=1e-5 if dtype == torch.float32 else 1e-2, atol=1e-5 if dtype == torch.float32 else 5e-2)
    a1, a0 = p1['concat'][0]['gaussian'], p0['concat'][0]['gaussian']
    assert torch.equal(a1['logvar'], a0['logvar'])
    close(a1['mu_mean'], a0['mu_mean'], rtol=1e-5, atol=1e-6 if dtype == torch.float32 else 1e-3)
    if dtype == torch.float32:
        close(g1, g0, rtol=1e-5, atol=2e-6)
    else:
        close(g1, g0, rtol=2e-2, atol=2e-2)


def test_concat_row_kernel_shard_invariant_philox():
    """In-kernel draws of the row kernel are keyed by the global row: a shard run with its row offset draws what the
    whole batch drew for those rows, both children."""
    from vae_amd import functional as F_
    B, L, K = 1000, 32, 64
    torch.manual_seed(2)
    logits = torch.randn(B, 2 * L + K, device=DEV)
    specs = (('gauss', L, 1e-6, None), ('gumbel', K, 0.8))
    seeds = ((5, 1), (5, 2))
    full = F_.concat_leaves(logits, specs, True, (None, None), seeds, row0=0)
    half = F_.concat_leaves(logits[B // 2:], specs, True, (None, None), seeds, row0=B // 2)
    assert torch.equal(half[0], full[0][B // 2:]) and torch.equal(half[2], full[2][B // 2:])


def test_concat_fused_philox_matches_unfused_streams():
    """Without injected noise both paths draw one (seed, offset) per stochastic child in order, so a
    re-seeded run of either gives the same sample."""
    import vae_amd
    from vae_amd.reparameterizers import ConcatReparameterizer
    m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'], cfg(continuous_size=64, discrete_size=16))
    m.train()
    logits = torch.randn(33, 80, device=DEV)
    vae_amd.rng.manual_seed(5)
    z1, _ = m(logits)
    vae_amd.rng.manual_seed(5)
    z2, _ = m(logits)
    z3, _ = m(logits)
    assert torch.equal(z1, z2) and not torch.equal(z2, z3)


class AffineFlow(torch.nn.Module):
    """Stand-in for the host project's flow network (reference flows/ is out of scope)."""

    def __init__(self, F):
        super().__init__()
        self.log_s = torch.nn.Parameter(torch.linspace(-0.5, 0.5, F))
        self.t = torch.nn.Parameter(torch.linspace(-1, 1, F))

    def forward(self, x):
        return x * torch.exp(self.log_s) + self.t, self.log_s.sum().expand(x.shape[0], 1)

    def sample(self, num_samples, noise):
        return (noise - self.t) * torch.exp(-self.log_s)


def test_sequential_gauss_flow_vs_oracle():
    from vae_amd.reparameterizers import SequentialReparameterizer
    torch.manual_seed(1)
    B, L = 64, 4
    c = cfg(continuous_size=2 * L, flow_builder=lambda conf: AffineFlow(conf['continuous_size']))
    m = SequentialReparameterizer(['isotropic_gaussian', 'flow'], c).to(DEV)
    m.train()
    assert m.input_size == 2 * L and m.output_size == 2 * L
    logits = torch.randn(B, 2 * L)
    noise = torch.randn(B, L)
    lg = logits.to(DEV).requires_grad_(True)
    z, params = m(lg, noise=[noise.to(DEV), None])
    assert set(params) == {'logits', 'sequential'} and len(params['sequential']) == 2
    assert set(params['sequential'][1]) == {'recon_x', 'logdet', 'logits'}
    kl = m.kl(params)
    # oracle: same chain on the CPU with the same weights
    mc = SequentialReparameterizer(['isotropic_gaussian', 'flow'], c)
    mc.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    lc = logits.clone().requires_grad_(True)
    z0, p0 = O.gauss_forward(lc, noise)
    proj, flow = mc.reparameterizers[1][0], mc.reparameterizers[1][1].flow
    u, logdet = flow(proj(z0))
    k0 = O.gauss_kl(p0) + O.flow_kl({'recon_x': u, 'logdet': logdet})
    close(z, u, rtol=1e-5, atol=1e-5)
    close(kl, k0, rtol=1e-5, atol=1e-4)
    (g1,) = torch.autograd.grad(z.sum() + kl.sum(), lg)
    (g0,) = torch.autograd.grad(u.sum() + k0.sum(), lc)
    close(g1, g0, rtol=1e-4, atol=1e-5)
    assert m.prior(5).shape == (5, 2 * L)


def test_sequential_golden_kl_split(golden):
    """The reference's own Gaussian->MAF run: its flow part of the KL from its recon_x / logdet."""
    from vae_amd import functional as F_
    g = golden('sequential')['sequential_gauss_flow']
    p = g['params']['sequential']
    kl_flow = F_.flow_kl(p[1]['recon_x'].to(DEV), p[1]['logdet'].to(DEV))
    kl_gauss = F_.gauss_kl(p[0]['gaussian']['mu'].to(DEV), p[0]['gaussian']['logvar'].to(DEV))
    close(kl_gauss + kl_flow, g['kl'], rtol=1e-5, atol=1e-4)


def test_mixture_constructs_and_runs():
    from vae_amd.reparameterizers import get_reparameterizer
    m = get_reparameterizer('mixture')(cfg())
    assert m.input_size == 13 and m.output_size == 9 and m.num_continuous_input == 8
    z, params = m(torch.randn(16, 13, device=DEV))
    assert z.shape == (16, 9) and set(params) == {'logits', 'z', 'gaussian', 'discrete'}
    assert m.kl(params).shape == (16,)


@pytest.mark.parametrize('train', [True, False])
def test_mixture_equals_concat_of_its_children(train):
    """Mixture(Gaussian, Gumbel) is the pair of the one-launch row kernel: with injected noise its sample, its
    per-child params and its KL equal the oracle's Concat restatement (the Gaussian part bit for bit), and d(logits)
    matches."""
    from vae_amd.reparameterizers import get_reparameterizer
    torch.manual_seed(12)
    B, L, K = 257, 64, 32
    m = get_reparameterizer('mixture')(cfg(continuous_size=2 * L, discrete_size=K)).train(train)
    logits, noise, U = torch.randn(B, 2 * L + K), torch.randn(B, L), torch.rand(B, K).clamp_(1e-3, 1 - 1e-3)
    w = torch.randn(B, L + K)
    lc = logits.clone().requires_grad_(True)
    z0, p0 = O.concat_forward(lc, [('isotropic_gaussian', 2 * L, {}), ('discrete', K, {'tau': 1.0})],
                              [noise if train else None, U], training=train)
    k0 = O.concat_kl(p0, ['isotropic_gaussian', 'discrete'])
    (g0,) = torch.autograd.grad((z0 * w).sum() + k0.sum(), lc)
    ld = logits.to(DEV).requires_grad_(True)
    z1, p1 = m(ld, noise=[noise.to(DEV) if train else None, U.to(DEV)])
    k1 = m.kl(p1)
    (g1,) = torch.autograd.grad((z1 * w.to(DEV)).sum() + k1.sum(), ld)
    assert set(p1) == {'logits', 'z', 'gaussian', 'discrete'} and p1['z'] is z1
    assert torch.equal(z1[:, :L].cpu(), z0[:, :L].detach())
    if train:
        close(z1[:, L:], z0[:, L:], rtol=1e-5, atol=1e-7)
    else:
        assert torch.equal(z1[:, L:].cpu() == 1, z0[:, L:] == 1)
    assert torch.equal(p1['gaussian']['logvar'].cpu(), p0['concat'][0]['gaussian']['logvar'].detach())
    assert torch.equal(p1['discrete']['z_hard'].cpu() == 1, p0['concat'][1]['discrete']['z_hard'] == 1)
    close(k1, k0, rtol=1e-5, atol=1e-4)
    close(g1, g0, rtol=1e-4, atol=1e-5)


def test_vrnn_golden(golden):
    from vae_amd import vrnn as V
    from vae_amd.reparameterizers import IsotropicGaussian
    g = golden('vrnn')['vrnn']
    L = 4
    c = cfg(continuous_size=2 * L)
    rep = IsotropicGaussian(c)
    rep.train()
    for s in g['steps']:
        lm = {'encoder_logits': s['enc_logits'].to(DEV), 'prior_logits': s['prior_logits'].to(DEV), 'x_features': None}
        z, params = V.reparameterize(rep, lm, c, noise=(s['noise_post'].to(DEV), s['noise_prior'].to(DEV)))
        assert set(z) == {'prior', 'posterior', 'x_features'} and set(params) == {'prior', 'posterior'}
        assert torch.equal(z['posterior'].cpu(), s['z_post'])
        close(z['prior'], s['z_prior'], rtol=1e-6, atol=1e-6)
        close(params['prior']['gaussian']['logvar'], s['params']['prior']['gaussian']['logvar'], rtol=1e-6, atol=1e-7)
        close(lm['prior_logits'], s['params']['prior']['logits'], rtol=1e-6, atol=1e-7)   # clamped in place of the map
        close(V.kld(rep, params, {'use_prior_kl': False}), s['kld'], rtol=1e-5, atol=1e-4)
        close(V.kld(rep, params, {'use_prior_kl': True}), s['kld_with_prior_kl'], rtol=1e-5, atol=1e-4)
    agg = V.loss_function_aggregate([{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in lt.items()}
                                     for lt in g['losses']])
    for k, v in g['mean_map'].items():
        if torch.is_tensor(v):
            close(agg[k], v, rtol=1e-6, atol=1e-7)
        else:
            assert agg[k] == pytest.approx(v)


def test_vrnn_step_logvars_are_produced_on_first_read():
    """The fused VRNN step writes neither logvar tensor: kld() uses the launch's own KLs and leaves them pending;
    reading them gives the reference's values and a gradient path into the logits."""
    from vae_amd import vrnn as V
    from vae_amd.lazy import LazyParams
    from vae_amd.reparameterizers import IsotropicGaussian
    torch.manual_seed(8)
    B, L = 37, 12
    c = cfg(continuous_size=2 * L)
    rep = IsotropicGaussian(c).train()
    enc, pri = torch.randn(B, 2 * L), torch.randn(B, 2 * L)
    nq, npr = torch.randn(B, L), torch.randn(B, L)
    ed, pd = enc.to(DEV).requires_grad_(True), pri.to(DEV).requires_grad_(True)
    lm = {'encoder_logits': ed, 'prior_logits': pd, 'x_features': None}
    z, params = V.reparameterize(rep, lm, c, noise=(nq.to(DEV), npr.to(DEV)))
    gq, gp = params['posterior']['gaussian'], params['prior']['gaussian']
    assert isinstance(gq, LazyParams) and gq.is_pending('logvar') and gp.is_pending('logvar')
    kl = V.kld(rep, params, {'use_prior_kl': True})
    assert gq.is_pending('logvar') and gp.is_pending('logvar')          # the KLs did not need them
    z0, p0 = O.vrnn_reparameterize(enc, pri, nq, npr)
    close(kl, O.vrnn_kld(p0, use_prior_kl=True), rtol=1e-5, atol=1e-4)
    assert torch.equal(gq['logvar'].cpu(), p0['posterior']['gaussian']['logvar'])
    close(gp['logvar'], p0['prior']['gaussian']['logvar'], rtol=1e-6, atol=1e-7)
    assert not gq.is_pending('logvar')
    (g,) = torch.autograd.grad(gq['logvar'].sum() + gp['logvar'].sum(), [ed], allow_unused=True)
    assert torch.equal(g[:, L:], torch.ones(B, L, device=DEV)) and float(g[:, :L].abs().max()) == 0.0
    # a KL against some other prior dict is recomputed, not taken from the cache
    other = {'gaussian': {'mu': gp['mu'].detach() + 1.0, 'logvar': gp['logvar'].detach()}}
    k2 = rep.kl(params['posterior'], other)
    assert not torch.allclose(k2, rep.kl(params['posterior'], params['prior']))


class _Anneal(object):
    """The stateful annealer of tests/golden/make_golden.py:make_vrnn_loss (advances once per compute_kl_beta call)."""

    def __init__(self):
        self.calls = 0
        self.seen = []

    def __call__(self, kl_beta_list):
        self.calls += 1
        res = [v * min(1.0, self.calls / 8.0) for v in kl_beta_list]
        self.seen.append(res[0])
        return res


def _vrnn_host(c, additive=False, own_nll=None):
    """A host model as INTEGRATION.md wires it: the mixin first, an nn.Module behind it."""
    from vae_amd import vrnn as V
    from vae_amd.reparameterizers import IsotropicGaussian

    class Host(V.AdditiveVRNNMixin if additive else V.VRNNMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.config = c
            self.reparameterizer = IsotropicGaussian(c)
            self.kl_annealer = _Anneal()

    if own_nll is not None:
        Host.nll = own_nll
    return Host().train()


def _golden_vrnn_steps(g, model, leaves_to=None):
    recons, xs, params, leaves = [], [], [], []
    for s in g['steps']:
        enc, pri, xhat, recon = (s[k].to(DEV).requires_grad_(True) for k in ('enc_logits', 'prior_logits',
                                                                               'xhat_logits', 'recon'))
        _, p = model.reparameterize({'encoder_logits': enc, 'prior_logits': pri, 'x_features': None},
                                    noise=(s['noise_post'].to(DEV), s['noise_prior'].to(DEV)))
        _, q = model.reparameterizer(xhat, noise=s['noise_xhat'].to(DEV))
        p['posterior']['q_z_given_xhat'] = q                       # vrnn.py:745-746
        close(model.mut_info(p, enc.shape[0]), s['mut_info'], rtol=1e-5, atol=1e-6)
        close(model.kld(p), s['kld'], rtol=1e-5, atol=1e-4)
        leaves += [enc, pri, xhat, recon]
        recons.append(recon)
        xs.append(s['x'].to(DEV))
        params.append(p)
    return recons, xs, params, leaves


@pytest.mark.parametrize('name', ['vrnnloss_vrnn', 'vrnnloss_additive'])
def test_vrnn_loss_function_golden_with_mut_info_prior_kl_and_annealer(golden, name):
    """The reference's VRNN.loss_function / AdditiveVRNN.loss_function run by the reference itself (T = 4) with
    use_prior_kl, continuous_mut_info > 0 under a clamp that bites and a kl_annealer that advances once per step
    (vrnn.py:751-765,805-829; abstract_vae.py:319-330,360-363,377): the drop-in mixin through the step-table launch."""
    g = golden('vrnnloss')[name]
    additive = name.endswith('additive')
    c = dict(g['config'], cuda=True)
    model = _vrnn_host(c, additive=additive)
    recons, xs, params, leaves = _golden_vrnn_steps(g, model)
    out = model.loss_function(recons, xs[0] if additive else xs, params)
    ref = g['result']
    assert set(out) == set(ref) and out['count'] == ref['count']
    assert model.kl_annealer.seen == pytest.approx(g['kl_betas'], rel=0, abs=0)       # one call per step, in order
    assert out['kl_beta_scalar'] == pytest.approx(ref['kl_beta_scalar'], rel=1e-12)
    for k in ('loss', 'elbo', 'loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'additional_loss_mean', 'mut_info_mean'):
        close(out[k], ref[k], rtol=1e-5, atol=1e-4)
    grads = torch.autograd.grad(out['loss_mean'], leaves, allow_unused=True)
    for i, s in enumerate(g['steps']):
        for j, k in enumerate(('enc_logits', 'prior_logits', 'xhat_logits', 'recon')):
            mine, want = grads[4 * i + j], s['grad_' + k]
            if want is None:
                assert mine is None or float(mine.abs().max()) == 0.0
            else:
                close(mine, want, rtol=1e-4, atol=1e-7)
    # a second call goes on from where the annealer stands (calls 5..8), as the reference's would
    model.loss_function([r.detach() for r in recons], xs[0] if additive else xs, params)
    assert model.kl_annealer.seen[4:] == [0.5 * min(1.0, n / 8.0) for n in range(5, 9)]


def test_vrnn_loss_function_own_nll_is_honoured(golden):
    """abstract_vae.py:344 calls self.nll: a host class that brings its own nll gets the per-step loop with THAT nll
    (here twice the Bernoulli NLL), not the fused launch's."""
    g = golden('vrnnloss')['vrnnloss_vrnn']
    c = dict(g['config'], cuda=True)

    def twice(self, x, recon_x, nll_type):
        from vae_amd import functional as F_
        return 2.0 * F_.nll(x, recon_x, nll_type)

    model = _vrnn_host(c, own_nll=twice)
    assert model._own_nll()
    recons, xs, params, _ = _golden_vrnn_steps(g, model)
    out = model.loss_function(recons, xs, params)
    plain = _vrnn_host(c)
    assert not plain._own_nll()
    r2, x2, p2, _ = _golden_vrnn_steps(g, plain)
    ref = plain.loss_function(r2, x2, p2)
    close(out['nll_mean'], 2.0 * ref['nll_mean'], rtol=1e-5, atol=1e-5)
    close(out['mut_info_mean'], ref['mut_info_mean'], rtol=1e-6, atol=1e-6)
    close(out['loss'] - out['elbo'], ref['loss'] - ref['elbo'], rtol=1e-4, atol=1e-4)   # (beta - 1) kld - mi: nll-free
    assert out['kl_beta_scalar'] == pytest.approx(ref['kl_beta_scalar'])


def test_elbo_mixin_own_nll_is_honoured():
    """ELBOMixin.loss_function with an overridden nll (K = 1 and K = 3): the override's values reach the loss dict."""
    from vae_amd.elbo import ELBOMixin
    from vae_amd.reparameterizers import IsotropicGaussian
    from vae_amd import functional as F_
    torch.manual_seed(21)
    B, L = 40, 8
    c = cfg(continuous_size=2 * L, kl_beta=0.3)
    dec = torch.nn.Linear(L, 784).to(DEV)

    class Host(ELBOMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.config = c
            self.reparameterizer = IsotropicGaussian(c)
            self.calls = 0

        def decode(self, z):
            return dec(z)

        def nll(self, x, recon_x, nll_type):
            self.calls += 1
            return 3.0 * F_.nll(x, recon_x, nll_type) + 1.0

    m = Host().train()
    x = (torch.rand(B, 784, device=DEV) < 0.2).float()
    logits = torch.randn(B, 2 * L, device=DEV)
    noise = torch.randn(B, L, device=DEV)
    z, params = m.reparameterizer(logits, noise=noise)
    recon = dec(z)
    out = m.loss_function(recon, x, params)
    assert m.calls == 1
    base = F_.nll(x, recon, 'bernoulli')
    kld = m.kld(params)
    ref = O.loss_function((3.0 * base + 1.0).cpu(), kld.cpu(), kl_beta=0.3)
    for k in ('loss', 'elbo', 'loss_mean', 'nll_mean', 'kld_mean'):
        close(out[k], ref[k], rtol=1e-5, atol=1e-3)
    m.calls = 0
    out3 = m.loss_function(recon, x, params, K=3)
    assert m.calls == 3 and out3['loss'].shape == (B,)


def test_vrnn_loss_per_step_beta_T64_vs_per_step_oracle():
    """T = 64 x B = 256 with a different KL weight at every step, a mutual-information vector and the prior KL:
    the step-table launch against the per-step oracle loop, forward and gradients (incl. d kld, d mi)."""
    from vae_amd import functional as F_
    torch.manual_seed(31)
    T, B, D = 64, 256, 784
    betas = [0.01 * (t + 1) for t in range(T)]
    xs = [(torch.rand(B, D) < 0.2).float() for _ in range(T)]
    recons = [torch.randn(B, D) for _ in range(T)]
    kld, mi = torch.rand(T * B) * 20, torch.randn(T * B)
    r0 = [r.clone().requires_grad_(True) for r in recons]
    k0, m0 = kld.clone().requires_grad_(True), mi.clone().requires_grad_(True)
    steps0 = [O.loss_function(O.nll(xs[t], r0[t], 'bernoulli'), k0[t * B:(t + 1) * B], betas[t], mut_info=m0[t * B:(t + 1) * B])
              for t in range(T)]
    ref = O.vrnn_loss(steps0)
    w = torch.randn(B)
    g0 = torch.autograd.grad(ref['loss_mean'] + (ref['loss'] * w).sum() / B, [r0[5], k0, m0])
    r1 = [r.to(DEV).requires_grad_(True) for r in recons]
    k1, m1 = kld.to(DEV).requires_grad_(True), mi.to(DEV).requires_grad_(True)
    _, _, _, means, la, ea = F_.nll_steps_elbo([x.to(DEV) for x in xs], r1, 'bernoulli', kld=k1, mi=m1, kl_beta=betas)
    close(la, ref['loss'], rtol=1e-5, atol=1e-3)
    close(ea, ref['elbo'], rtol=1e-5, atol=1e-3)
    for i, k in enumerate(('loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'additional_loss_mean', 'mut_info_mean')):
        close(means[i], ref[k], rtol=1e-5, atol=1e-4)
    g1 = torch.autograd.grad(means[0] + (la * w.to(DEV)).sum() / B, [r1[5], k1, m1])
    for a, b in zip(g1, g0):
        close(a, b, rtol=1e-4, atol=1e-9)


def test_vrnn_time_batched_loss_vs_per_step_oracle():
    """T steps scored in one launch over [T*B] rows == the reference's per-step loop + _mean_map."""
    from vae_amd import vrnn as V
    from vae_amd.reparameterizers import IsotropicGaussian
    torch.manual_seed(3)
    T, B, L = 5, 32, 8
    c = cfg(continuous_size=2 * L, use_prior_kl=True, kl_beta=0.5)
    rep = IsotropicGaussian(c)
    rep.train()
    xs = [(torch.rand(B, 1, 28, 28) < 0.2).float() for _ in range(T)]
    recons = [torch.randn(B, 1, 28, 28) for _ in range(T)]
    encs = [torch.randn(B, 2 * L) for _ in range(T)]
    pris = [torch.randn(B, 2 * L) for _ in range(T)]
    n1 = [torch.randn(B, L) for _ in range(T)]
    n2 = [torch.randn(B, L) for _ in range(T)]
    steps0, params1 = [], []
    for t in range(T):
        _, p0 = O.vrnn_reparameterize(encs[t], pris[t], n1[t], n2[t])
        steps0.append(O.loss_function(O.nll(xs[t], recons[t], 'bernoulli'), O.vrnn_kld(p0, True), 0.5))
        _, p1 = V.reparameterize(rep, {'encoder_logits': encs[t].to(DEV), 'prior_logits': pris[t].to(DEV),
                                       'x_features': None}, c, noise=(n1[t].to(DEV), n2[t].to(DEV)))
        params1.append(p1)
    ref = O.vrnn_loss(steps0)
    out = V.loss_function(rep, c, [r.to(DEV) for r in recons], [x.to(DEV) for x in xs], params1)
    assert set(out) == set(ref)
    for k, v in ref.items():
        if torch.is_tensor(v):
            close(out[k], v, rtol=1e-5, atol=1e-3)
        else:
            assert out[k] == pytest.approx(v)


def test_vrnn_loss_T64_full_step_vs_per_step_oracle():
    """BASELINE configs[4] at its sequence length: T = 64 steps of B_step = 1024, 1x28x28 Bernoulli, the per-step
    reconstructions and targets read in place (vae_nll_steps_elbo_fwd/bwd: pointer tables, no torch.cat) against the
    reference's per-step loop + _add_loss_map / _mean_map (vrnn.py:767-829) on the CPU, forward and gradients."""
    from vae_amd import vrnn as V
    from vae_amd.reparameterizers import IsotropicGaussian
    torch.manual_seed(8)
    T, B, L = 64, 1024, 16
    c = cfg(continuous_size=2 * L, use_prior_kl=False, kl_beta=0.7)
    rep = IsotropicGaussian(c)
    rep.train()
    xs = [(torch.rand(B, 1, 28, 28) < 0.2).float() for _ in range(T)]
    recons = [torch.randn(B, 1, 28, 28) for _ in range(T)]
    encs = [torch.randn(B, 2 * L) for _ in range(T)]
    pris = [torch.randn(B, 2 * L) for _ in range(T)]
    n1 = [torch.randn(B, L) for _ in range(T)]
    n2 = [torch.randn(B, L) for _ in range(T)]
    w = torch.randn(B)
    r0 = [r.clone().requires_grad_(True) for r in recons]
    e0 = [e.clone().requires_grad_(True) for e in encs]
    steps0 = []
    for t in range(T):
        _, p0 = O.vrnn_reparameterize(e0[t], pris[t], n1[t], n2[t])
        steps0.append(O.loss_function(O.nll(xs[t], r0[t], 'bernoulli'), O.vrnn_kld(p0, False), 0.7))
    ref = O.vrnn_loss(steps0)
    g0 = torch.autograd.grad(ref['loss_mean'] + (ref['loss'] * w).sum() / B, [r0[0], r0[T - 1], e0[3]])
    r1 = [r.to(DEV).requires_grad_(True) for r in recons]
    e1 = [e.to(DEV).requires_grad_(True) for e in encs]
    params1 = []
    for t in range(T):
        _, p1 = V.reparameterize(rep, {'encoder_logits': e1[t], 'prior_logits': pris[t].to(DEV), 'x_features': None}, c,
                                 noise=(n1[t].to(DEV), n2[t].to(DEV)))
        params1.append(p1)
    out = V.loss_function(rep, c, r1, [x.to(DEV) for x in xs], params1)
    assert set(out) == set(ref) and out['count'] == T and out['loss'].shape == (B,)
    for k, v in ref.items():
        if torch.is_tensor(v):
            close(out[k], v, rtol=1e-5, atol=1e-3)
    g1 = torch.autograd.grad(out['loss_mean'] + (out['loss'] * w.to(DEV)).sum() / B, [r1[0], r1[T - 1], e1[3]])
    for a, b in zip(g1, g0):
        close(a, b, rtol=1e-4, atol=2e-7 * max(1.0, float(b.abs().max()) * 1e4))


def test_vrnn_loss_additive_scores_every_step_on_the_last_reconstruction():
    """additive_vrnn.py:94-96: one target, the LAST reconstruction at every step -- the same tensor T times in the
    pointer tables; its gradient is the sum over the steps."""
    from vae_amd import vrnn as V
    from vae_amd.reparameterizers import IsotropicGaussian
    torch.manual_seed(9)
    T, B, L = 4, 50, 6
    c = cfg(continuous_size=2 * L, use_prior_kl=False, kl_beta=1.0)
    rep = IsotropicGaussian(c)
    rep.train()
    x = (torch.rand(B, 1, 28, 28) < 0.2).float()
    recons = [torch.randn(B, 1, 28, 28) for _ in range(T)]
    encs, pris = [torch.randn(B, 2 * L) for _ in range(T)], [torch.randn(B, 2 * L) for _ in range(T)]
    n1, n2 = [torch.randn(B, L) for _ in range(T)], [torch.randn(B, L) for _ in range(T)]
    rl0 = recons[-1].clone().requires_grad_(True)
    steps0 = []
    for t in range(T):
        _, p0 = O.vrnn_reparameterize(encs[t], pris[t], n1[t], n2[t])
        steps0.append(O.loss_function(O.nll(x, rl0, 'bernoulli'), O.vrnn_kld(p0, False), 1.0))
    ref = O.vrnn_loss(steps0)
    (g0,) = torch.autograd.grad(ref['loss_mean'], rl0)
    r1 = [r.to(DEV).requires_grad_(True) for r in recons]
    params1 = []
    for t in range(T):
        _, p1 = V.reparameterize(rep, {'encoder_logits': encs[t].to(DEV), 'prior_logits': pris[t].to(DEV),
                                       'x_features': None}, c, noise=(n1[t].to(DEV), n2[t].to(DEV)))
        params1.append(p1)
    out = V.loss_function(rep, c, r1, x.to(DEV), params1, additive=True)
    for k, v in ref.items():
        if torch.is_tensor(v):
            close(out[k], v, rtol=1e-5, atol=1e-3)
    g1 = torch.autograd.grad(out['loss_mean'], r1, allow_unused=True)
    assert all(g is None or float(g.abs().max()) == 0.0 for g in g1[:-1])
    close(g1[-1], g0, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize('nll_type', ['bernoulli', 'gaussian', 'disc_logistic'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_nll_steps_equals_concatenated_launch(nll_type, dtype):
    """vae_nll_steps_elbo_fwd/bwd over T separate tensors == vae_nll_elbo_fwd/bwd over their concatenation, bit for
    bit (same kernels, another row source), incl. a ragged B and a row length that takes the scalar path."""
    from vae_amd import functional as F_
    torch.manual_seed(12)
    for T, B, D in ((3, 37, 75), (7, 5, 784), (2, 3, 12288)):
        R = D if nll_type == 'bernoulli' else 2 * D
        xs = [torch.rand(B, D).to(dtype).to(DEV) for _ in range(T)]
        rs = [(torch.randn(B, R) * 1.5).to(dtype).to(DEV).requires_grad_(True) for _ in range(T)]
        kld = torch.rand(T * B).to(dtype).to(DEV).requires_grad_(True)
        w = torch.randn(T * B, device=DEV).to(dtype)
        nll, loss, elbo, means, la, ea = F_.nll_steps_elbo(xs, rs, nll_type, kld=kld, kl_beta=0.6)
        g = torch.autograd.grad((loss.float() * w.float()).sum() + means[0] + 2 * means[1], rs + [kld])
        rc = torch.cat([r.detach() for r in rs], 0).requires_grad_(True)
        kc = kld.detach().clone().requires_grad_(True)
        n2, l2, e2, m2 = F_.nll_elbo(torch.cat(xs, 0), rc, nll_type, kld=kc, kl_beta=0.6)
        g2 = torch.autograd.grad((l2.float() * w.float()).sum() + m2[0] + 2 * m2[1], [rc, kc])
        assert torch.equal(nll, n2) and torch.equal(loss, l2) and torch.equal(elbo, e2) and torch.equal(means, m2)
        assert torch.equal(torch.cat(g[:T], 0), g2[0]) and torch.equal(g[T], g2[1])
        acc_l = loss.view(T, B)[0].clone()
        for t in range(1, T):
            acc_l = acc_l + loss.view(T, B)[t]
        assert torch.equal(la, acc_l / T)


@pytest.mark.parametrize('B,L', [(5, 3), (64, 128), (1000, 36), (257, 512)])
@pytest.mark.parametrize('train', [True, False])
def test_vrnn_fused_pair_backward_vs_oracle(B, L, train):
    """The fused per-step kernel (both draws + clamp + both KLs): values and all input gradients."""
    from vae_amd import vrnn as V
    from vae_amd.reparameterizers import IsotropicGaussian
    torch.manual_seed(B + L)
    c = cfg(continuous_size=2 * L)
    rep = IsotropicGaussian(c)
    rep.train(train)
    enc, pri = torch.randn(B, 2 * L), torch.randn(B, 2 * L)
    n1, n2 = torch.randn(B, L), torch.randn(B, L)
    w1, w2, w3 = torch.randn(B, L), torch.randn(B, L), torch.randn(B)

    ec, pc = enc.clone().requires_grad_(True), pri.clone().requires_grad_(True)
    z0, p0 = O.vrnn_reparameterize(ec, pc, n1, n2, training=train)
    k0 = O.vrnn_kld(p0, True)
    obj0 = (z0['posterior'] * w1).sum() + (z0['prior'] * w2).sum() + (k0 * w3).sum() \
        + p0['prior']['gaussian']['logvar_mean'] + 2 * p0['posterior']['gaussian']['mu_mean'] \
        + (p0['prior']['gaussian']['logvar'] * w1).sum() + (p0['prior']['logits'] * torch.cat([w2, w1], -1)).sum()
    g0 = torch.autograd.grad(obj0, [ec, pc])

    ed, pd = enc.to(DEV).requires_grad_(True), pri.to(DEV).requires_grad_(True)
    lm = {'encoder_logits': ed, 'prior_logits': pd, 'x_features': None}
    z1, p1 = V.reparameterize(rep, lm, c, noise=(n1.to(DEV), n2.to(DEV)))
    k1 = V.kld(rep, p1, {'use_prior_kl': True})
    w1d, w2d, w3d = w1.to(DEV), w2.to(DEV), w3.to(DEV)
    obj1 = (z1['posterior'] * w1d).sum() + (z1['prior'] * w2d).sum() + (k1 * w3d).sum() \
        + p1['prior']['gaussian']['logvar_mean'] + 2 * p1['posterior']['gaussian']['mu_mean'] \
        + (p1['prior']['gaussian']['logvar'] * w1d).sum() + (p1['prior']['logits'] * torch.cat([w2d, w1d], -1)).sum()
    g1 = torch.autograd.grad(obj1, [ed, pd])
    assert torch.equal(z1['posterior'].detach().cpu(), z0['posterior'].detach())
    close(z1['prior'], z0['prior'], rtol=1e-6, atol=1e-6)
    close(k1, k0, rtol=1e-5, atol=1e-5 * L)
    close(V.kld(rep, p1, {'use_prior_kl': False}), O.vrnn_kld(p0, False), rtol=1e-5, atol=1e-5 * L)
    close(g1[0], g0[0], rtol=1e-4, atol=1e-5)
    close(g1[1], g0[1], rtol=1e-4, atol=1e-5)
    # kl() against a DIFFERENT prior must not reuse the fused pair value
    other = {'gaussian': {'mu': torch.zeros(B, L, device=DEV), 'logvar': torch.ones(B, L, device=DEV)}}
    close(rep.kl(p1['posterior'], other), O.gauss_kl(p0['posterior']), rtol=1e-5, atol=1e-5 * L)


@pytest.mark.parametrize('name', ['cond_train_nopkl', 'cond_train_pkl', 'cond_eval_nopkl', 'cond_eval_pkl'])
def test_conditioned_prior_golden(golden, name):
    """ConditionedPriorMixin (one fused pair launch) against ClassConditionedPriorVAE.reparameterize / kld run by
    the reference: params schema, bit-exact draws, KL and the gradients into the encoder logits and the label
    projector."""
    from vae_amd.conditioned import ConditionedPriorMixin
    from vae_amd.reparameterizers import IsotropicGaussian
    g = golden('cond')[name]
    L = g['z'].shape[1]

    class Model(ConditionedPriorMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.config = cfg(continuous_size=2 * L, output_size=g['num_classes'], use_prior_kl=g['use_prior_kl'])
            self.reparameterizer = IsotropicGaussian(self.config)
            self.prior_mlp = torch.nn.Linear(g['num_classes'], 2 * L)

    m = Model().to(DEV)
    with torch.no_grad():
        m.prior_mlp.weight.copy_(g['mlp_weight'])
        m.prior_mlp.bias.copy_(g['mlp_bias'])
    m.train(g['training'])
    logits = g['logits'].to(DEV).requires_grad_(True)
    z, params = m.reparameterize(logits, g['labels'].to(DEV),
                                 noise=(g['noise_post'].to(DEV), g['noise_prior'].to(DEV)))
    assert set(params) == set(g['params'])
    for side in ('prior', 'posterior'):
        assert set(params[side]) == set(g['params'][side])
        assert set(params[side]['gaussian']) == set(g['params'][side]['gaussian'])
    assert torch.equal(z.detach().cpu(), g['z'])
    close(params['prior']['logits'], g['params']['prior']['logits'], rtol=1e-6, atol=1e-7)
    close(params['prior']['gaussian']['logvar'], g['params']['prior']['gaussian']['logvar'], rtol=1e-6, atol=1e-7)
    kl = m.kld(params)
    close(kl, g['kld'], rtol=1e-5, atol=1e-5)
    gl, gw, gb = torch.autograd.grad((z * g['w'].to(DEV)).sum() + kl.sum(), [logits, m.prior_mlp.weight, m.prior_mlp.bias])
    close(gl, g['grad_logits'], rtol=1e-4, atol=1e-5)
    close(gw, g['grad_mlp_weight'], rtol=1e-4, atol=1e-5)
    close(gb, g['grad_mlp_bias'], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ config 4 end to end
def ref_cfg(**kw):
    """The reference's own config keys for a Gaussian -> Flow chain (tests/golden/make_golden.py: base_config)."""
    c = {'continuous_size': 8, 'discrete_size': 5, 'half': False, 'cuda': True, 'epochs': 10,
         'continuous_mut_info': 0, 'discrete_mut_info': 0, 'kl_beta': 1.0, 'use_prior_kl': False,
         'flow_type': 'maf', 'latent_size': 16, 'memory_read_steps': 1, 'reparam_type': 'isotropic_gaussian'}
    c.update(kw)
    return c


def test_flow_reparameterizer_builds_from_reference_config(golden):
    """FlowReparameterizer(config) with flow_type / continuous_size / latent_size only (flow.py:20-49), the
    reference's own state_dict loaded strictly, its eval-mode forward and "KL" reproduced."""
    from vae_amd.reparameterizers import FlowReparameterizer
    g = golden('flow')['flow_forward']
    m = FlowReparameterizer(ref_cfg(continuous_size=6, latent_size=16))
    m.load_state_dict(g['state_dict'], strict=True)
    m = m.to(DEV).eval()
    f, p = m(g['x'].to(DEV))
    assert set(p) == {'recon_x', 'logdet'} and p['logdet'].shape == g['logdet'].shape
    close(f, g['flow'], rtol=1e-5, atol=1e-5)
    close(p['logdet'], g['logdet'], rtol=1e-5, atol=1e-5)
    close(m.kl(p), g['kl'], rtol=1e-5, atol=1e-5)
    assert m.input_size == 6 and m.output_size == 6
    for ft in ('maf_split', 'realnvp', 'glow', 'maf_split_glow'):
        FlowReparameterizer(ref_cfg(continuous_size=6, flow_type=ft, dense_normalization='none'))
    with pytest.raises(ValueError):
        FlowReparameterizer(ref_cfg(flow_type='planar'))


def test_sequential_gauss_flow_golden_end_to_end(golden):
    """SequentialReparameterizer(['isotropic_gaussian', 'flow'], cfg) from the reference's config alone, the
    reference's state_dict (Gaussian -> Linear -> 5 x (MADE, BatchNormFlow, Reverse)) loaded strictly, its noise
    injected: z, every params entry, kl and d/d logits of (z.sum() + kl.sum()) against the reference's own run."""
    from vae_amd.reparameterizers import SequentialReparameterizer
    g = golden('sequential')['sequential_gauss_flow']
    m = SequentialReparameterizer(['isotropic_gaussian', 'flow'], ref_cfg(continuous_size=8, latent_size=16))
    assert m.input_size == g['input_size'] and m.output_size == g['output_size']
    m.load_state_dict(g['state_dict'], strict=True)
    m = m.to(DEV).train(True)
    logits = g['logits'].to(DEV).requires_grad_(True)
    z, params = m(logits, noise=[g['noise'].to(DEV), None])
    assert set(params) == {'logits', 'sequential'} and len(params['sequential']) == 2
    want0, want1 = g['params']['sequential']
    got0, got1 = params['sequential']
    assert set(got0) == set(want0) and set(got1) == set(want1) and set(got0['gaussian']) == set(want0['gaussian'])
    assert torch.equal(got0['z'].detach().cpu(), want0['z'])                       # the Gaussian draw: bit-exact
    assert torch.equal(got0['gaussian']['logvar'].detach().cpu(), want0['gaussian']['logvar'])
    close(got0['gaussian']['mu_mean'], want0['gaussian']['mu_mean'], rtol=1e-5, atol=1e-6)
    close(got1['recon_x'], want1['recon_x'], rtol=1e-5, atol=1e-5)
    close(got1['logdet'], want1['logdet'], rtol=1e-5, atol=1e-5)
    close(z, g['z'], rtol=1e-5, atol=1e-5)
    kl = m.kl(params)
    close(kl, g['kl'], rtol=1e-5, atol=1e-4)
    (gl,) = torch.autograd.grad(z.sum() + kl.sum(), logits)
    close(gl, g['grad_logits'], rtol=1e-4, atol=1e-4)


def _maf_oracle(sd, prefix, x, nblocks, training):
    """CPU restatement of FlowSequential(5 x (MADE(tanh), BatchNormFlow, Reverse)) from a state_dict, through the
    oracle's tails (flows/__init__.py:176-178, 233-257, 403-406, 555-557)."""
    import torch.nn.functional as Fn
    logdet = torch.zeros(x.shape[0], 1, dtype=x.dtype)
    for b in range(nblocks):
        p = '%s%d.' % (prefix, 3 * b)
        lin = lambda h, name: Fn.linear(h, sd[p + name + '.linear.weight'] * sd[p + name + '.mask'],
                                        sd[p + name + '.linear.bias'])
        ma = lin(torch.tanh(lin(torch.tanh(lin(x, 'joiner')), 'trunk.1')), 'trunk.3')
        u, ld = O.made_tail(x, *ma.chunk(2, 1))
        q = '%s%d.' % (prefix, 3 * b + 1)
        y, ld2, _, _ = O.bnflow(u, sd[q + 'log_gamma'], sd[q + 'beta'], sd[q + 'running_mean'], sd[q + 'running_var'],
                                training)
        x, logdet = O.reverse_columns(y), logdet + ld + ld2
    return x, logdet


def test_config4_chain_full_size():
    """BASELINE configs[3] at size: B = 8192, Gaussian(L = 128) -> dense -> MAF(F = 256, 5 blocks), 3x64x64
    discretized-logistic NLL, through the drop-in classes built from the reference's config keys alone.
    Checked: the chain's sample, logdet and KL against the CPU oracle with the same weights and noise; the loss
    dict's size-independent properties (elbo == nll + kld per sample, means == means of the vectors); and shard
    invariance of every per-sample output in eval mode (batch statistics make training-mode shards differ by
    construction: vae_amd.dist reduces them, tests/test_gpu_dist.py).

    Tolerances: the chain comparison is rtol 1e-4 (atol 1e-3 on z, 1e-2 on the logdet, 5e-2 on a KL of magnitude
    ~1e3), not 1e-5, because fifteen stock masked GEMMs with K = 256 (hipBLASLt on the GPU, MKL on the CPU: different
    summation orders, then amplified by exp(-a) and the batch-norm division) sit between the kernels; that looseness is
    the GEMMs', not the kernels'.  The kernels' own share is held tighter elsewhere: the Gaussian KL of this very chain
    to torch.equal (end of this test), the Gaussian draw bit for bit (test_sequential_gauss_flow_golden_end_to_end),
    the MADE / BatchNormFlow / Reverse tails at rtol 1e-5 with their inputs given (tests/test_gpu_flowtail.py, also
    the folded MADE + moments launch this chain takes in training), the Flow KL at 1e-5 (tests/test_gpu_flow.py)."""
    from vae_amd.reparameterizers import SequentialReparameterizer
    from vae_amd import elbo as E
    torch.manual_seed(4)
    B, L = 8192, 128
    c = ref_cfg(continuous_size=2 * L, latent_size=256, nll_type='disc_logistic')
    m = SequentialReparameterizer(['isotropic_gaussian', 'flow'], c).to(DEV).train(True)
    logits, noise = torch.randn(B, 2 * L), torch.randn(B, L)
    lg = logits.to(DEV).requires_grad_(True)
    z, params = m(lg, noise=[noise.to(DEV), None])
    kl = m.kl(params)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    z0, p0 = O.gauss_forward(logits, noise)
    h = torch.nn.functional.linear(z0, sd['reparameterizers.1.0.weight'], sd['reparameterizers.1.0.bias'])
    u0, ld0 = _maf_oracle(sd, 'reparameterizers.1.1.flow.', h, 5, True)
    k0 = O.gauss_kl(p0) + O.flow_kl({'recon_x': u0, 'logdet': ld0})
    close(z, u0, rtol=1e-4, atol=1e-3)
    close(params['sequential'][1]['logdet'], ld0, rtol=1e-4, atol=1e-2)
    close(kl, k0, rtol=1e-4, atol=5e-2)
    # the loss over 3x64x64 images
    x = (torch.floor(torch.rand(B, 3, 64, 64, device=DEV) * 256) / 255.0).clamp(0, 1)
    recon = torch.randn(B, 6, 64, 64, device=DEV, requires_grad=True)
    out = E.loss_function(recon, x, kl, 'disc_logistic', kl_beta=1.0)
    assert out['loss'].shape == (B,) and out['elbo'].shape == (B,)
    nll = E.nll(x, recon, 'disc_logistic')
    close(out['elbo'], nll + kl, rtol=1e-6, atol=1e-2)
    close(out['elbo_mean'], out['elbo'].double().mean().float(), rtol=1e-5)
    close(out['kld_mean'], kl.double().mean().float(), rtol=1e-5)
    close(out['nll_mean'], nll.double().mean().float(), rtol=1e-5)
    (out['loss_mean'] + z.mean()).backward()
    assert bool(torch.isfinite(lg.grad).all()) and bool(torch.isfinite(recon.grad).all())
    # eval mode: rows are independent -> two half-batch calls equal the full call, row for row
    m.eval()
    with torch.no_grad():
        zf, pf = m(lg)
        klf = m.kl(pf)
        za, pa = m(lg[:B // 2])
        zb, pb = m(lg[B // 2:])
        # the stock GEMMs in between pick their kernels by M: to rounding, not bit for bit (the kernels' own outputs
        # are: the Gaussian KL below, tests/test_gpu_fullsize.py)
        close(torch.cat([za, zb]), zf, rtol=1e-5, atol=1e-5)
        # the Flow "KL" is a cross-batch mean: only the per-sample part kl + logdet - mean is shardable
        ga = m.reparameterizers[0].kl(pa['sequential'][0])
        gf = m.reparameterizers[0].kl(pf['sequential'][0])
        assert torch.equal(ga, gf[:B // 2])
        close(torch.cat([pa['sequential'][1]['logdet'], pb['sequential'][1]['logdet']]),
              pf['sequential'][1]['logdet'], rtol=1e-5, atol=1e-4)
        assert klf.shape == (B,)
