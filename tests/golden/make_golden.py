"""Generate tests/golden/*.pt by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py

/root/reference is imported read-only as the package `models.vae` (its own absolute imports
need that name).  Two of its dependencies are absent from the image (`helpers`, `tree`); they
are registered as in-memory modules holding only the trivial utilities the hot path calls
(zeros_like / ones_like / same_type / eps / nan_check_and_break, a Linear for the Sequential
projection).  Nothing from the reference is copied: the fixtures are inputs, the injected
noise, and the outputs the reference computed (data only).  The NLL lives in the absent
`helpers.distributions` and is therefore NOT in any fixture (parity unpinned, see
oracle/cpu_ref.py).

The GPU box has no /root/reference; only the .pt files travel.
"""
import os
import sys
import types
import importlib.util

import numpy as np
import torch
import torch.nn as nn
import torch.distributions as D

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
EPS_TOL = 1e-6  # helpers.utils.eps(half=False): value assumed, helpers is un-vendored


def install_reference():
    D.Distribution.set_default_validate_args(False)  # the reference predates validation-by-default

    helpers = types.ModuleType('helpers')
    utils = types.ModuleType('helpers.utils')
    layers = types.ModuleType('helpers.layers')
    dists = types.ModuleType('helpers.distributions')
    tree = types.ModuleType('tree')

    utils.zeros_like = torch.zeros_like
    utils.ones_like = torch.ones_like
    utils.eps = lambda half=False: 1e-4 if half else EPS_TOL
    utils.nan_check_and_break = lambda t, name='': None
    utils.float_type = lambda cuda: torch.FloatTensor
    utils.long_type = lambda cuda: torch.LongTensor
    utils.same_type = lambda half, cuda: torch.HalfTensor if half else torch.FloatTensor
    utils.add_noise_to_imgs = lambda x: x
    utils.number_of_parameters = lambda m: sum(p.numel() for p in m.parameters())
    utils.one_hot_np = None

    def get_encoder(name='dense', **config):
        in_size = config['input_shape'][0]
        return lambda output_size: nn.Linear(in_size, output_size)

    layers.get_encoder = get_encoder
    layers.get_decoder = get_encoder
    layers.str_to_activ_module = lambda s: {'tanh': nn.Tanh, 'relu': nn.ReLU}[s]

    class EMA(nn.Module):
        def __init__(self, decay=0.999):
            super().__init__()

        def forward(self, x):
            return x

    layers.EMA = EMA
    for n in ('flatten_layers', 'LinearWarmupWithCosineAnnealing', 'GatedDense', 'SineLinear', 'SNLinear',
              'Identity', 'View'):
        setattr(layers, n, None)
    # stock dense block of the un-vendored helpers: a plain layer_fn(in, out) stands in (the coupling NETS are not
    # on the path; only what the reference does with their outputs is pinned)
    layers.BasicDenseBlock = lambda i, o, layer_fn=nn.Linear, normalization_str='none', activation_str='relu': \
        layer_fn(i, o)

    def _absent(*a, **k):
        raise RuntimeError('helpers.distributions is un-vendored: NLL is not part of any fixture')

    dists.nll = _absent
    dists.nll_activation = _absent
    dists.nll_has_variance = lambda s: False
    tree.map_structure = lambda f, *s: f(*s)

    helpers.utils, helpers.layers, helpers.distributions = utils, layers, dists
    sys.modules.update({'helpers': helpers, 'helpers.utils': utils, 'helpers.layers': layers,
                        'helpers.distributions': dists, 'tree': tree})

    models = types.ModuleType('models')
    models.__path__ = []
    sys.modules['models'] = models
    spec = importlib.util.spec_from_file_location('models.vae', os.path.join(REF, '__init__.py'),
                                                  submodule_search_locations=[REF])
    vae = importlib.util.module_from_spec(spec)
    sys.modules['models.vae'] = vae
    spec.loader.exec_module(vae)
    models.vae = vae
    return vae


def detach_tree(o):
    if torch.is_tensor(o):
        return o.detach().clone()
    if isinstance(o, dict):
        return {k: detach_tree(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [detach_tree(v) for v in o]
    if hasattr(o, 'item'):  # numpy scalar (tau after annealing)
        return float(o)
    return o


def grads_of(z, kl, logits, w):
    (g,) = torch.autograd.grad((z * w).sum() + kl.sum(), logits, allow_unused=True)
    return None if g is None else g.detach().clone()


def base_config(**kw):
    c = {'continuous_size': 8, 'discrete_size': 5, 'half': False, 'cuda': False, 'epochs': 10,
         'continuous_mut_info': 0, 'discrete_mut_info': 0, 'kl_beta': 1.0, 'use_prior_kl': False,
         'flow_type': 'maf', 'latent_size': 16, 'memory_read_steps': 1, 'reparam_type': 'isotropic_gaussian'}
    c.update(kw)
    return c


def make_gaussian(vae, out):
    R = vae.reparameterizers
    for L in (4, 32):
        for train in (True, False):
            torch.manual_seed(100 + L)
            B = 8
            logits = torch.randn(B, 2 * L, requires_grad=True)
            prior_logits = torch.randn(B, 2 * L)
            w = torch.randn(B, L)
            m = R.IsotropicGaussian(base_config(continuous_size=2 * L))
            m.train(train)
            torch.manual_seed(7)
            noise = torch.zeros(B, L).normal_()
            torch.manual_seed(7)
            z, params = m(logits)
            kl = m.kl(params)
            # a second parameter set as the prior (VRNN / class-conditioned use): sigma made positive
            pl = torch.cat([prior_logits[:, :L], torch.sigmoid(prior_logits[:, L:])], -1).requires_grad_(True)
            _, pparams = m(pl)
            klp = m.kl(params, pparams)
            g = grads_of(z, kl, logits, w)
            gp_q, gp_p = torch.autograd.grad(klp.sum(), [logits, pl])
            out['gauss_L%d_%s' % (L, 'train' if train else 'eval')] = detach_tree({
                'logits': logits, 'noise': noise, 'w': w, 'z': z, 'params': params, 'kl': kl,
                'prior_logits': pl, 'prior_params': pparams, 'kl_prior': klp, 'grad_logits': g,
                'grad_klp_logits': gp_q, 'grad_klp_prior_logits': gp_p, 'training': train, 'eps_tol': EPS_TOL})
    # 3-D logits [B, T, 2L] (slice -1, kl [B, T])
    torch.manual_seed(5)
    logits = torch.randn(4, 3, 16)
    m = R.IsotropicGaussian(base_config(continuous_size=16))
    torch.manual_seed(9)
    noise = torch.zeros(4, 3, 8).normal_()
    torch.manual_seed(9)
    z, params = m(logits)
    out['gauss_3d'] = detach_tree({'logits': logits, 'noise': noise, 'z': z, 'params': params, 'kl': m.kl(params)})
    # 4-D logits [B, 2C, H, W]: split on the CHANNEL dimension (:108-114), kl over W only -> [B, C, H]
    torch.manual_seed(6)
    logits = torch.randn(3, 6, 5, 8, requires_grad=True)
    m = R.IsotropicGaussian(base_config(continuous_size=6))
    torch.manual_seed(10)
    noise = torch.zeros(3, 3, 5, 8).normal_()
    torch.manual_seed(10)
    z, params = m(logits)
    kl = m.kl(params)
    w, wk = torch.randn(3, 3, 5, 8), torch.randn(3, 3, 5)
    (g,) = torch.autograd.grad((z * w).sum() + (kl * wk).sum(), logits)
    out['gauss_4d'] = detach_tree({'logits': logits, 'noise': noise, 'z': z, 'params': params, 'kl': kl, 'w': w, 'wk': wk,
                                   'grad_logits': g})


def make_gumbel(vae, out):
    R = vae.reparameterizers
    for K in (5, 32):
        for train in (True, False):
            for tau in (1.0, 0.5):
                torch.manual_seed(200 + K)
                B = 8
                logits = torch.randn(B, K)
                # rows 0/1: exact ties after noise are forced below through U; rows 2/3: near ties
                logits = logits.requires_grad_(True)
                prior_logits = torch.randn(B, K, requires_grad=True)
                w = torch.randn(B, K)
                m = R.GumbelSoftmax(base_config(discrete_size=K))
                m.train(train)
                m.tau = tau
                torch.manual_seed(11)
                U = torch.rand(B, K)
                torch.manual_seed(11)
                z, params = m(logits)
                kl0 = m.kl(params)
                _, pparams = m(prior_logits)
                klp = m.kl(params, pparams)
                g, gprior = torch.autograd.grad((z * w).sum() + klp.sum(), [logits, prior_logits])
                hard = params['discrete']['z_hard'].detach()
                out['gumbel_K%d_%s_tau%g' % (K, 'train' if train else 'eval', tau)] = detach_tree({
                    'logits': logits, 'U': U, 'w': w, 'tau': tau, 'z': z, 'params': params, 'kl_none': kl0,
                    'prior_logits': prior_logits, 'prior_log_q_z': pparams['discrete']['log_q_z'],
                    'kl_prior': klp, 'grad_logits': g, 'grad_prior_logits': gprior,
                    'hard_index': hard.argmax(-1), 'hard_count': (hard == 1).sum(-1), 'training': train})
    # exact ties: identical logits and identical U in several columns -> multi-hot z_hard
    torch.manual_seed(3)
    K = 8
    logits = torch.randn(6, K)
    logits[:, 5] = logits[:, 2]
    logits[0, 7] = logits[0, 2]
    big = logits.max() + 3.0
    logits[:, 2] = big
    logits[:, 5] = big
    logits[0, 7] = big
    m = R.GumbelSoftmax(base_config(discrete_size=K))
    m.train(False)

    class FixedRand(object):
        """Make torch.rand return a chosen draw for one call (noise injection without touching the reference)."""

        def __init__(self, U):
            self.U, self.orig = U, torch.rand

        def __enter__(self):
            torch.rand = lambda *a, **k: self.U.clone()

        def __exit__(self, *a):
            torch.rand = self.orig

    U = torch.rand(6, K)
    U[:, 5] = U[:, 2]
    U[0, 7] = U[0, 2]
    with FixedRand(U):
        z, params = m(logits)
    hard = params['discrete']['z_hard']
    assert int((hard == 1).sum(-1).max()) >= 2
    out['gumbel_ties'] = detach_tree({'logits': logits, 'U': U, 'tau': 1.0, 'z': z, 'params': params,
                                      'hard_count': (hard == 1).sum(-1), 'training': False})
    # annealing: iteration / tau evolution over 25 training forwards (gumbel.py:91-108, :334)
    m = R.GumbelSoftmax(base_config(discrete_size=4))
    m.train(True)
    m.iteration = 99990
    taus = []
    for _ in range(25):
        m(torch.zeros(2, 4))
        taus.append(float(m.tau))
    out['gumbel_anneal'] = {'start_iteration': 99990, 'taus': taus, 'end_iteration': m.iteration}


def make_bernoulli(vae, out):
    R = vae.reparameterizers
    for K in (5, 32):
        for train in (True, False):
            for tau in (1.0, 0.3):
                torch.manual_seed(300 + K)
                B = 8
                logits = (torch.randn(B, K) * 2).requires_grad_(True)
                prior_logits = torch.randn(B, K, requires_grad=True)
                w = torch.randn(B, K)
                m = R.Bernoulli(base_config(discrete_size=K))
                m.train(train)
                m.tau = tau
                torch.manual_seed(13)
                U = torch.rand(B, K)
                torch.manual_seed(13)
                z, params = m(logits)
                kl0 = m.kl(params)
                klp = m.kl(params, {'discrete': {'logits': prior_logits}})
                g = grads_of(z, kl0, logits, w)
                gq, gp = torch.autograd.grad(klp.sum(), [logits, prior_logits])
                out['bern_K%d_%s_tau%g' % (K, 'train' if train else 'eval', tau)] = detach_tree({
                    'logits': logits, 'U': U, 'w': w, 'tau': tau, 'z': z, 'params': params, 'kl_none': kl0,
                    'prior_logits': prior_logits, 'kl_prior': klp, 'grad_logits': g,
                    'grad_klp_logits': gq, 'grad_klp_prior_logits': gp, 'training': train})


def make_beta(vae, out):
    R = vae.reparameterizers
    for L in (4, 32):
        torch.manual_seed(400 + L)
        B = 8
        logits = (torch.randn(B, 2 * L) * 2).requires_grad_(True)
        prior_logits = torch.randn(B, 2 * L, requires_grad=True)
        m = R.Beta(base_config(continuous_size=2 * L))
        m.train(True)
        z, params = m(logits)
        _, pparams = m(prior_logits)
        kl0 = m.kl(params)
        klp = m.kl(params, pparams)
        ll = m.log_likelihood(z.detach(), params)
        (g0,) = torch.autograd.grad(kl0.sum() + params['beta']['conc1_mean'] + 2 * params['beta']['conc2_mean'],
                                    logits, retain_graph=True)
        gq, gp = torch.autograd.grad(klp.sum(), [logits, prior_logits])
        out['beta_L%d' % L] = detach_tree({
            'logits': logits, 'z_drawn': z, 'params': params, 'kl_none': kl0, 'prior_logits': prior_logits,
            'kl_prior': klp, 'log_likelihood': ll, 'grad_kl0_means_logits': g0, 'grad_klp_logits': gq,
            'grad_klp_prior_logits': gp})


def make_flow_kl(vae, out):
    R = vae.reparameterizers
    torch.manual_seed(500)
    m = R.FlowReparameterizer(base_config(continuous_size=6, latent_size=16))
    # 2-D: logdet [B,1] -> the [B,B] broadcast of SURVEY Q6
    u = torch.randn(8, 6, requires_grad=True)
    logdet = torch.randn(8, 1, requires_grad=True)
    kl = m.kl({'recon_x': u, 'logdet': logdet})
    gu, gl = torch.autograd.grad((kl * torch.arange(1., 9.)).sum(), [u, logdet])
    out['flowkl_2d'] = detach_tree({'u': u, 'logdet': logdet, 'kl': kl, 'w': torch.arange(1., 9.),
                                    'grad_u': gu, 'grad_logdet': gl})
    # 3-D: logdet [B,R]
    u = torch.randn(4, 3, 6, requires_grad=True)
    logdet = torch.randn(4, 3, requires_grad=True)
    kl = m.kl({'recon_x': u, 'logdet': logdet})
    gu, gl = torch.autograd.grad((kl * torch.arange(1., 5.)).sum(), [u, logdet])
    out['flowkl_3d'] = detach_tree({'u': u, 'logdet': logdet, 'kl': kl, 'w': torch.arange(1., 5.),
                                    'grad_u': gu, 'grad_logdet': gl})
    # the full module forward through the real MAF flow (weights saved) for the Sequential drop-in test
    m.eval()
    x = torch.randn(8, 6)
    f, p = m(x)
    out['flow_forward'] = detach_tree({'x': x, 'flow': f, 'logdet': p['logdet'], 'kl': m.kl(p),
                                       'state_dict': {k: v for k, v in m.state_dict().items()}})


def make_concat(vae, out):
    R = vae.reparameterizers
    for train in (True, False):
        torch.manual_seed(600)
        B, L, K = 8, 4, 5
        cfg = base_config(continuous_size=2 * L, discrete_size=K)
        m = R.ConcatReparameterizer(['isotropic_gaussian', 'discrete'], cfg)
        m.train(train)
        logits = torch.randn(B, 2 * L + K, requires_grad=True)
        w = torch.randn(B, L + K)
        torch.manual_seed(17)
        noise = torch.zeros(B, L).normal_() if train else None
        U = torch.rand(B, K)
        torch.manual_seed(17)
        z, params = m(logits)
        kl = m.kl(params)
        g = grads_of(z, kl, logits, w)
        out['concat_%s' % ('train' if train else 'eval')] = detach_tree({
            'logits': logits, 'noise': noise, 'U': U, 'w': w, 'z': z, 'params': params, 'kl': kl,
            'grad_logits': g, 'input_size': m.input_size, 'output_size': m.output_size, 'training': train,
            'scalars': m.get_reparameterizer_scalars()})


def make_sequential(vae, out):
    R = vae.reparameterizers
    torch.manual_seed(700)
    B, L, Fdim = 8, 4, 6
    cfg = base_config(continuous_size=2 * L, latent_size=16)
    # Gaussian -> Flow: the flow reads config['continuous_size'] too, so the reference's own config
    # couples F to 2L; use F = 2L = 8
    m = R.SequentialReparameterizer(['isotropic_gaussian', 'flow'], cfg)
    m.train(True)
    logits = torch.randn(B, 2 * L, requires_grad=True)
    torch.manual_seed(19)
    noise = torch.zeros(B, L).normal_()
    torch.manual_seed(19)
    z, params = m(logits)
    kl = m.kl(params)
    (g,) = torch.autograd.grad(z.sum() + kl.sum(), logits)
    out['sequential_gauss_flow'] = detach_tree({
        'logits': logits, 'noise': noise, 'z': z, 'params': params, 'kl': kl, 'grad_logits': g,
        'input_size': m.input_size, 'output_size': m.output_size,
        'state_dict': {k: v for k, v in m.state_dict().items()}})


def make_elbo(vae, out):
    """AbstractVAE.loss_function called unbound on a stub: pins the combine (abstract_vae.py:356-388)."""
    AV = vae.abstract_vae.AbstractVAE if hasattr(vae, 'abstract_vae') else sys.modules['models.vae.abstract_vae'].AbstractVAE
    torch.manual_seed(800)
    B = 16
    nll = torch.rand(B) * 100
    kld = torch.rand(B) * 10
    extra = {'a': torch.rand(B), 'b': torch.rand(B)}
    for beta, with_extra in ((1.0, False), (0.25, True)):
        class Stub(object):
            training = False
            config = {'nll_type': 'bernoulli', 'kl_beta': beta, 'half': True, 'cuda': False}
            kl_annealer = None
            nll = lambda self, x, r, t: nll  # noqa: E731
            kld = lambda self, p: kld  # noqa: E731
            mut_info = lambda self, p, b: torch.zeros(b)  # noqa: E731
            compute_kl_beta = AV.compute_kl_beta
        res = AV.loss_function(Stub(), None, torch.zeros(B, 1), {}, **(extra if with_extra else {}))
        out['elbo_beta%g_%s' % (beta, 'extra' if with_extra else 'plain')] = detach_tree({
            'nll': nll, 'kld': kld, 'extra': extra if with_extra else {}, 'kl_beta': beta, 'result': res})


def make_vrnn(vae, out):
    """VRNN.reparameterize / kld / _add_loss_map / _mean_map on stubs (vrnn.py:395-447, :723-829)."""
    V = sys.modules['models.vae.vrnn'].VRNN
    R = vae.reparameterizers
    torch.manual_seed(900)
    B, L, T = 8, 4, 4
    cfg = base_config(continuous_size=2 * L, reparam_type='isotropic_gaussian')
    rep = R.IsotropicGaussian(cfg)
    rep.train(True)

    class Stub(object):
        config = cfg
        reparameterizer = rep
        _clamp_variance = V._clamp_variance
        kld = V.kld
    steps, losses = [], []
    for t in range(T):
        enc = torch.randn(B, 2 * L)
        pri = torch.randn(B, 2 * L)
        torch.manual_seed(1000 + t)
        n_post = torch.zeros(B, L).normal_()
        n_prior = torch.zeros(B, L).normal_()
        torch.manual_seed(1000 + t)
        z, params = V.reparameterize(Stub(), {'encoder_logits': enc, 'prior_logits': pri.clone(), 'x_features': None})
        cfg['use_prior_kl'] = False
        kl = V.kld(Stub(), params)
        cfg['use_prior_kl'] = True
        kl2 = V.kld(Stub(), params)
        cfg['use_prior_kl'] = False
        steps.append(detach_tree({'enc_logits': enc, 'prior_logits': pri, 'noise_post': n_post, 'noise_prior': n_prior,
                                  'z_post': z['posterior'], 'z_prior': z['prior'], 'params': params, 'kld': kl,
                                  'kld_with_prior_kl': kl2}))
        losses.append({'loss': torch.rand(B), 'elbo': torch.rand(B), 'loss_mean': torch.rand(()),
                       'kl_beta_scalar': 1.0})
    agg = None
    for lt in losses:
        agg = V._add_loss_map({k: (v.clone() if torch.is_tensor(v) else v) for k, v in lt.items()}, agg)
    agg = V._mean_map(agg)
    out['vrnn'] = {'steps': steps, 'losses': detach_tree(losses), 'mean_map': detach_tree(agg)}


def make_multi(vae, out):
    """The draw loops: loss_function(K=3) in training (abstract_vae.py:347-354), MSGVAE.reparameterize
    (msg.py:55-69) and likelihood (abstract_vae.py:281-317), each called unbound on a stub.  The NLL is the
    stub's own (Bernoulli via torch.distributions; helpers is absent): these fixtures pin the draw order, the
    K-average and the log-sum-exp normalisation, not an NLL definition."""
    AV = sys.modules['models.vae.abstract_vae'].AbstractVAE
    MSG = sys.modules['models.vae.msg'].MSGVAE
    R = vae.reparameterizers
    B, L, Dx = 8, 4, 12
    cfg = base_config(continuous_size=2 * L, nll_type='bernoulli', max_time_steps=3)
    torch.manual_seed(1100)
    dec = nn.Linear(L, Dx)
    enc = nn.Linear(Dx, 2 * L)

    class Stub(object):
        training = True
        config = cfg
        kl_annealer = None
        reparameterizer = R.IsotropicGaussian(cfg)
        compute_kl_beta = AV.compute_kl_beta
        loss_function = AV.loss_function

        def reparameterize(self, logits, labels=None, force=False):
            return self.reparameterizer(logits, force=force)

        def decode(self, z):
            return dec(z)

        def encode(self, x):
            return enc(x)

        def nll(self, x, recon_x, nll_type):
            return -torch.sum(D.Bernoulli(logits=recon_x).log_prob(x), -1)

        def kld(self, p):
            return self.reparameterizer.kl(p)

        def mut_info(self, p, b):
            return torch.zeros(b)

    st = Stub()
    st.reparameterizer.train(True)
    x = (torch.rand(B, Dx) < 0.3).float()
    logits = torch.randn(B, 2 * L, requires_grad=True)
    K = 3
    torch.manual_seed(21)
    noises = [torch.zeros(B, L).normal_() for _ in range(K)]
    torch.manual_seed(21)
    z0, params = st.reparameterize(logits)
    res = AV.loss_function(st, st.decode(z0), x, params, K=K)
    g_logits, g_w, g_b = torch.autograd.grad(res['loss_mean'], [logits, dec.weight, dec.bias])
    out['multi_lossK3'] = detach_tree({
        'x': x, 'logits': logits, 'noises': noises, 'K': K, 'dec_weight': dec.weight, 'dec_bias': dec.bias,
        'result': res, 'grad_logits': g_logits, 'grad_dec_weight': g_w, 'grad_dec_bias': g_b})

    # MSGVAE.reparameterize + kld (last parameter set)
    torch.manual_seed(22)
    noises = [torch.zeros(B, L).normal_() for _ in range(3)]
    torch.manual_seed(22)
    z_list, p_list = MSG.reparameterize(st, logits)
    kl = MSG.kld(st, p_list)
    w = torch.randn(3, B, L)
    (g,) = torch.autograd.grad(sum((z * wi).sum() for z, wi in zip(z_list, w)) + kl.sum(), logits)
    out['multi_msg'] = detach_tree({'logits': logits, 'noises': noises, 'z_list': z_list, 'params_last': p_list[-1],
                                    'kl': kl, 'w': w, 'grad_logits': g})

    # likelihood: model in eval (loss_function takes no extra draws), reparameterizer left in TRAIN mode in one
    # case so the draws are stochastic, and in eval in the other (z = mu: every value identical)
    Bl, Kl = 4, 8
    xb = (torch.rand(Bl, Dx) < 0.3).float()
    for mode in (True, False):
        st.training = False
        st.reparameterizer.train(mode)
        torch.manual_seed(23)
        noises = [torch.zeros(Bl, L).normal_() for _ in range(Bl * (Kl // Bl))]
        torch.manual_seed(23)
        val = AV.likelihood(st, [(xb, torch.zeros(Bl).long())], K=Kl)
        out['multi_likelihood_%s' % ('draws' if mode else 'mean')] = detach_tree({
            'x': xb, 'K': Kl, 'noises': noises if mode else None, 'enc_weight': enc.weight, 'enc_bias': enc.bias,
            'dec_weight': dec.weight, 'dec_bias': dec.bias, 'value': val, 'reparam_training': mode})


def make_mi(vae, out):
    """mutual_info() of the leaf reparameterizers (isotropic_gaussian.py:142-156, gumbel.py:143-168,
    beta.py:138-152): params of x and of the re-encoded reconstruction, both produced by the reference."""
    R = vae.reparameterizers
    B = 8
    torch.manual_seed(1200)
    for name, cls, size_key, width, scale_key in (('gauss', R.IsotropicGaussian, 'continuous_size', 8, 'continuous_mut_info'),
                                                  ('gumbel', R.GumbelSoftmax, 'discrete_size', 6, 'discrete_mut_info'),
                                                  ('beta', R.Beta, 'continuous_size', 8, 'continuous_mut_info')):
        cfg = base_config(**{size_key: width, scale_key: 0.7})
        m = cls(cfg)
        m.train(True)
        la = torch.randn(B, width, requires_grad=True)
        lb = torch.randn(B, width, requires_grad=True)
        if name == 'gauss':   # scales must be positive for Normal-vs-Normal KL to be finite
            la = torch.cat([la[:, :4], la[:, 4:].abs() + 0.1], -1).detach().requires_grad_(True)
            lb = torch.cat([lb[:, :4], lb[:, 4:].abs() + 0.1], -1).detach().requires_grad_(True)
        _, pa = m(la)
        _, pb = m(lb)
        mi = m.mutual_info({**pa, 'q_z_given_xhat': pb})
        ga, gb = torch.autograd.grad((mi * torch.arange(1., B + 1)).sum(), [la, lb])
        out['mi_%s' % name] = detach_tree({'logits': la, 'logits_xhat': lb, 'scale': 0.7, 'mi': mi,
                                           'w': torch.arange(1., B + 1), 'grad_logits': ga, 'grad_logits_xhat': gb})


def make_conditioned(vae, out):
    """ClassConditionedPriorVAE.reparameterize / kld / _process_labels unbound on a stub (conditioned.py:44-147)."""
    C = sys.modules['models.vae.conditioned'].ClassConditionedPriorVAE
    R = vae.reparameterizers
    B, L, NC = 8, 4, 5
    for train in (True, False):
        for use_prior_kl in (False, True):
            torch.manual_seed(1300)
            cfg = base_config(continuous_size=2 * L, output_size=NC, use_prior_kl=use_prior_kl)
            mlp = nn.Linear(NC, 2 * L)

            class Stub(object):
                config = cfg
                reparameterizer = R.IsotropicGaussian(cfg)
                prior_mlp = mlp
                _clamp_variance = C._clamp_variance
                _process_labels = C._process_labels

            st = Stub()
            st.reparameterizer.train(train)
            logits = torch.randn(B, 2 * L, requires_grad=True)
            labels = torch.randint(0, NC, (B,))
            torch.manual_seed(31)
            n_post, n_prior = torch.zeros(B, L).normal_(), torch.zeros(B, L).normal_()
            torch.manual_seed(31)
            z, params = C.reparameterize(st, logits, labels)
            kl = C.kld(st, params)
            w = torch.randn(B, L)
            g_logits, g_w, g_b = torch.autograd.grad((z * w).sum() + kl.sum(), [logits, mlp.weight, mlp.bias])
            out['cond_%s_%s' % ('train' if train else 'eval', 'pkl' if use_prior_kl else 'nopkl')] = detach_tree({
                'logits': logits, 'labels': labels, 'noise_post': n_post, 'noise_prior': n_prior, 'w': w,
                'mlp_weight': mlp.weight, 'mlp_bias': mlp.bias, 'z': z, 'params': params, 'kld': kl,
                'grad_logits': g_logits, 'grad_mlp_weight': g_w, 'grad_mlp_bias': g_b, 'training': train,
                'use_prior_kl': use_prior_kl, 'num_classes': NC})


def make_flowtail(vae, out):
    """flows/__init__.py layers run by the reference itself: MADE, MADESplit (+pre_exp_tanh), BatchNormFlow
    (train / eval), CouplingLayer, and whole MAF / MAF-split / RealNVP stacks (weights saved)."""
    FL = sys.modules['models.vae.flows']
    B, Fd, Hd = 8, 6, 16

    def sd(m):
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    def grads(outs, ws, inputs):
        obj = sum((o * w).sum() for o, w in zip(outs, ws))
        return [None if g is None else g.detach().clone() for g in torch.autograd.grad(obj, inputs, allow_unused=True)]

    torch.manual_seed(1400)
    x = torch.randn(B, Fd, requires_grad=True)
    wu, wl = torch.randn(B, Fd), torch.randn(B, 1)
    def tap(mod, store, key):   # record a stock sub-network's output (the tail's input)
        return mod.register_forward_hook(lambda _m, _i, o: store.__setitem__(key, o.detach().clone()))

    m = FL.MADE(Fd, Hd)
    taps = {}
    tap(m.trunk, taps, 'ma')
    u, ld = m(x)
    g = grads([u, ld], [wu, wl], [x, m.trunk[3].linear.weight, m.joiner.linear.bias])
    out['flowtail_made'] = detach_tree({'x': x, 'wu': wu, 'wl': wl, 'state_dict': sd(m), 'u': u, 'logdet': ld, **taps,
                                        'grad_x': g[0], 'grad_out_weight': g[1], 'grad_joiner_bias': g[2]})
    for tanh in (False, True):
        m = FL.MADESplit(Fd, Hd, pre_exp_tanh=tanh)
        taps = {}
        tap(m.s_trunk, taps, 'm')
        tap(m.t_trunk, taps, 'a')
        u, ld = m(x)
        g = grads([u, ld], [wu, wl], [x, m.t_trunk[3].linear.weight, m.s_trunk[3].linear.weight])
        out['flowtail_madesplit_%s' % ('tanh' if tanh else 'plain')] = detach_tree({
            'x': x, 'wu': wu, 'wl': wl, 'state_dict': sd(m), 'u': u, 'logdet': ld, 'pre_exp_tanh': tanh, **taps,
            'grad_x': g[0], 'grad_t_out_weight': g[1], 'grad_s_out_weight': g[2]})
    for train in (True, False):
        m = FL.BatchNormFlow(Fd, momentum=0.1)
        with torch.no_grad():
            m.log_gamma.normal_(0, 0.3)
            m.beta.normal_()
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2.0)
        before = sd(m)
        m.train(train)
        xs = (x * 1.7 + 0.4).detach().requires_grad_(True)
        y, ld = m(xs)
        g = grads([y, ld.expand(B, 1)], [wu, wl], [xs, m.log_gamma, m.beta])
        out['flowtail_bn_%s' % ('train' if train else 'eval')] = detach_tree({
            'x': xs, 'wu': wu, 'wl': wl, 'state_dict_before': before, 'state_dict_after': sd(m), 'y': y,
            'logdet': ld, 'training': train, 'momentum': 0.1, 'eps': 1e-5, 'grad_x': g[0],
            'grad_log_gamma': g[1], 'grad_beta': g[2]})
    mask = (torch.arange(0, Fd) % 2).float()
    m = FL.CouplingLayer(Fd, Hd, mask)
    taps = {}
    tap(m.scale_net, taps, 's_raw')
    tap(m.translate_net, taps, 't_raw')
    y, ld = m(x)
    g = grads([y, ld], [wu, wl], [x, m.scale_net[4].weight, m.translate_net[4].weight])
    out['flowtail_coupling'] = detach_tree({'x': x, 'wu': wu, 'wl': wl, 'mask': mask, 'state_dict': sd(m), 'y': y, **taps,
                                            'logdet': ld, 'grad_x': g[0], 'grad_s_out_weight': g[1],
                                            'grad_t_out_weight': g[2]})
    m = FL.AdditiveCouplingLayer(Fd, Hd, mask)
    taps = {}
    tap(m.net, taps, 't_raw')
    y, ld = m(x)
    g = grads([y], [wu], [x])
    out['flowtail_additive'] = detach_tree({'x': x, 'wu': wu, 'mask': mask, 'state_dict': sd(m), 'y': y, 'logdet': ld, **taps,
                                            'grad_x': g[0]})
    # whole stacks, training mode (batch statistics), two blocks each
    for name, build in (('maf', lambda: FL.build_maf_flow(Fd, Hd, num_blocks=2)),
                        ('mafsplit', lambda: FL.build_maf_split_flow(Fd, Hd, num_blocks=2)),
                        ('realnvp', lambda: FL.build_realnvp_flow(Fd, Hd, num_blocks=2))):
        torch.manual_seed(1410)
        m = build()
        for mod in m.modules():
            if isinstance(mod, FL.BatchNormFlow):
                with torch.no_grad():
                    mod.log_gamma.normal_(0, 0.2)
                    mod.beta.normal_(0, 0.5)
        before = sd(m)
        m.train(True)
        u, ld = m(x)
        params = [p for p in m.parameters()]
        g = grads([u, ld], [wu, wl], [x] + params)
        lp = m.log_probs(x)
        m.eval()
        ue, lde = m(x)
        out['flowtail_stack_%s' % name] = detach_tree({
            'x': x, 'wu': wu, 'wl': wl, 'state_dict_before': before, 'state_dict_after_train': sd(m), 'u': u,
            'logdet': ld, 'grad_x': g[0], 'grad_params': g[1:], 'param_names': [k for k, _ in m.named_parameters()],
            'log_probs_second_train_call': lp, 'u_eval': ue, 'logdet_eval': lde})


def make_glow(vae, out):
    """The glow-side layers of flows/__init__.py run by the reference itself: ActNorm (data-dependent first call),
    InvertibleMM, LUInvertibleMM, Shuffle, and two-block glow / maf_split_glow stacks (flows/__init__.py:274-389,
    612-625, 651-673), forward, inverse and gradients."""
    FL = sys.modules['models.vae.flows']
    B, Fd, Hd = 8, 6, 16

    def sd(m):
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    def grads(outs, ws, inputs):
        obj = sum((o * w).sum() for o, w in zip(outs, ws))
        return [None if g is None else g.detach().clone() for g in torch.autograd.grad(obj, inputs, allow_unused=True)]

    torch.manual_seed(1500)
    x = torch.randn(B, Fd, requires_grad=True)
    wu, wl = torch.randn(B, Fd), torch.randn(B, 1)

    m = FL.ActNorm(Fd)
    xs = (x * 1.7 + 0.4).detach().requires_grad_(True)
    y, ld = m(xs)                                   # first call: weight / bias from the batch
    after_init = sd(m)
    y2, ld2 = m(x)                                  # second call: the parameters as they are
    g = grads([y2, ld2], [wu, wl], [x, m.weight, m.bias])
    xi, ldi = m(y2.detach(), mode='inverse')
    out['glow_actnorm'] = detach_tree({'x_first': xs, 'y_first': y, 'logdet_first': ld, 'state_dict_after_init': after_init,
                                       'x': x, 'wu': wu, 'wl': wl, 'y': y2, 'logdet': ld2, 'grad_x': g[0],
                                       'grad_weight': g[1], 'grad_bias': g[2], 'x_inverse': xi, 'logdet_inverse': ldi})
    m = FL.InvertibleMM(Fd)
    y, ld = m(x)
    g = grads([y, ld], [wu, wl], [x, m.W])
    xi, ldi = m(y.detach(), mode='inverse')
    out['glow_invmm'] = detach_tree({'x': x, 'wu': wu, 'wl': wl, 'state_dict': sd(m), 'y': y, 'logdet': ld, 'grad_x': g[0],
                                     'grad_W': g[1], 'x_inverse': xi, 'logdet_inverse': ldi})
    m = FL.LUInvertibleMM(Fd)
    y, ld = m(x)
    g = grads([y, ld], [wu, wl], [x, m.L, m.U, m.log_S])
    xi, ldi = m(y.detach(), mode='inverse')
    out['glow_lumm'] = detach_tree({'x': x, 'wu': wu, 'wl': wl, 'state_dict': sd(m), 'P': m.P, 'sign_S': m.sign_S, 'y': y,
                                    'logdet': ld, 'grad_x': g[0], 'grad_L': g[1], 'grad_U': g[2], 'grad_log_S': g[3],
                                    'x_inverse': xi, 'logdet_inverse': ldi})
    m = FL.Shuffle(Fd)
    y, ld = m(x)
    xi, _ = m(y.detach(), mode='inverse')
    out['glow_shuffle'] = detach_tree({'x': x, 'perm': torch.from_numpy(np.asarray(m.perm)).long(),
                                       'inv_perm': torch.from_numpy(np.asarray(m.inv_perm)).long(), 'y': y, 'logdet': ld, 'x_inverse': xi})
    for name, build in (('glow', lambda: FL.build_glow_flow(Fd, Hd, num_blocks=2, activation_str='relu',
                                                            normalization_str='none', layer_modifier='none')),
                        ('mafsplitglow', lambda: FL.build_maf_split_glow_flow(Fd, Hd, num_blocks=2))):
        torch.manual_seed(1510)
        m = build()
        m.train(True)
        before = sd(m)
        u, ld = m(x)                                # ActNorm layers initialise themselves on this call
        after = sd(m)
        u2, ld2 = m(x)
        params = [p for p in m.parameters()]
        g = grads([u2, ld2], [wu, wl], [x] + params)
        out['glow_stack_%s' % name] = detach_tree({
            'x': x, 'wu': wu, 'wl': wl, 'state_dict_before': before, 'state_dict_after_first_call': after, 'u_first': u,
            'logdet_first': ld, 'u': u2, 'logdet': ld2, 'grad_x': g[0], 'grad_params': g[1:],
            'param_names': [k for k, _ in m.named_parameters()]})


def make_vrnn_loss(vae, out):
    """VRNN.loss_function / AdditiveVRNN.loss_function themselves (vrnn.py:805-829, additive_vrnn.py:78-104) on an
    un-constructed VRNN instance: T = 4 steps of AbstractVAE.loss_function with everything a step can carry switched
    on -- use_prior_kl, continuous_mut_info > 0 with a clamp that bites (vrnn.py:751-765, abstract_vae.py:473-488) and
    a STATEFUL kl_annealer (abstract_vae.py:319-330, :363: it advances once per step).  The NLL is the stub's own
    (Bernoulli via torch.distributions; helpers is absent): the fixture pins the per-step combine and the
    aggregation, not an NLL definition."""
    V = sys.modules['models.vae.vrnn'].VRNN
    AVR = sys.modules['models.vae.additive_vrnn'].AdditiveVRNN
    R = vae.reparameterizers
    B, L, T, Dx = 8, 4, 4, 12

    class Anneal(object):
        def __init__(self):
            self.calls = 0
            self.seen = []

        def __call__(self, kl_beta_list):
            self.calls += 1
            res = [v * min(1.0, self.calls / 8.0) for v in kl_beta_list]
            self.seen.append(res[0])
            return res

    for name, cls in (('vrnn', V), ('additive', AVR)):
        torch.manual_seed(1600)
        cfg = base_config(continuous_size=2 * L, reparam_type='isotropic_gaussian', use_prior_kl=True, kl_beta=0.5,
                          continuous_mut_info=0.7, mut_clamp_strategy='clamp', mut_clamp_value=1.5,
                          nll_type='bernoulli')
        rep = R.IsotropicGaussian(cfg)
        rep.train(True)
        st = cls.__new__(cls)
        nn.Module.__init__(st)
        st.config = cfg
        st.reparameterizer = rep
        st.kl_annealer = Anneal()
        st.nll = lambda x, recon_x, nll_type: -torch.sum(D.Bernoulli(logits=recon_x).log_prob(x), -1)
        steps, recons, xs, params = [], [], [], []
        for t in range(T):
            enc = torch.randn(B, 2 * L)
            enc = torch.cat([enc[:, :L], enc[:, L:].abs() + 0.5], -1).requires_grad_(True)
            pri = torch.randn(B, 2 * L, requires_grad=True)
            xhat = (enc.detach() + 0.4 * torch.randn(B, 2 * L)).requires_grad_(True)   # the re-encoded reconstruction
            recon = torch.randn(B, Dx, requires_grad=True)
            x_t = (torch.rand(B, Dx) < 0.3).float()
            torch.manual_seed(1700 + t)
            n_post = torch.zeros(B, L).normal_()
            n_prior = torch.zeros(B, L).normal_()
            n_xhat = torch.zeros(B, L).normal_()
            torch.manual_seed(1700 + t)
            _, p = cls.reparameterize(st, {'encoder_logits': enc, 'prior_logits': pri, 'x_features': None})
            _, q_xhat = rep(xhat)                                   # vrnn.py:745-746
            p['posterior']['q_z_given_xhat'] = q_xhat
            steps.append({'enc_logits': enc, 'prior_logits': pri, 'xhat_logits': xhat, 'noise_post': n_post,
                          'noise_prior': n_prior, 'noise_xhat': n_xhat, 'recon': recon, 'x': x_t,
                          'mut_info': cls.mut_info(st, p, B), 'kld': cls.kld(st, p)})
            recons.append(recon)
            xs.append(x_t)
            params.append(p)
        if name == 'additive':
            res = cls.loss_function(st, recons, xs[0], params)     # one target: every step scored on the last recon
        else:
            res = cls.loss_function(st, recons, xs, params)
        leaves = [s_[k] for s_ in steps for k in ('enc_logits', 'prior_logits', 'xhat_logits', 'recon')]
        grads = torch.autograd.grad(res['loss_mean'], leaves, allow_unused=True)
        for i, s_ in enumerate(steps):
            for j, k in enumerate(('enc_logits', 'prior_logits', 'xhat_logits', 'recon')):
                g = grads[4 * i + j]
                s_['grad_' + k] = None if g is None else g.detach().clone()
        out['vrnnloss_%s' % name] = detach_tree({
            'steps': steps, 'result': res, 'kl_betas': list(st.kl_annealer.seen), 'config': dict(cfg),
            'annealer': 'beta * min(1, calls / 8), calls counted from 1 over the life of the model'})


def main():
    vae = install_reference()
    out = {}
    make_gaussian(vae, out)
    make_gumbel(vae, out)
    make_bernoulli(vae, out)
    make_beta(vae, out)
    make_flow_kl(vae, out)
    make_concat(vae, out)
    make_sequential(vae, out)
    make_elbo(vae, out)
    make_vrnn(vae, out)
    make_multi(vae, out)
    make_mi(vae, out)
    make_conditioned(vae, out)
    make_flowtail(vae, out)
    make_glow(vae, out)
    make_vrnn_loss(vae, out)
    groups = {}
    for k, v in out.items():
        groups.setdefault(k.split('_')[0], {})[k] = v
    for g, d in groups.items():
        path = os.path.join(HERE, g + '.pt')
        torch.save(d, path)
        print('wrote %s (%d cases, %d bytes)' % (path, len(d), os.path.getsize(path)))


if __name__ == '__main__':
    main()
