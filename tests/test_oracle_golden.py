"""Pin oracle/cpu_ref.py against outputs of the reference itself (tests/golden/*.pt).

CPU only.  Where the op order is identical the comparison is bit-exact (torch.equal);
elsewhere rtol 1e-6.
"""
import torch
import pytest

from oracle import cpu_ref as O


def close(a, b, rtol=1e-6, atol=1e-7):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize('name', ['gauss_L4_train', 'gauss_L4_eval', 'gauss_L32_train', 'gauss_L32_eval', 'gauss_3d'])
def test_gaussian(golden, name):
    g = golden('gauss')[name]
    training = g.get('training', True)
    logits = g['logits'].clone().requires_grad_(True)
    z, params = O.gauss_forward(logits, g['noise'], training=training)
    assert torch.equal(z, g['z'])
    gp = g['params']['gaussian']
    assert torch.equal(params['gaussian']['mu'], gp['mu'])
    assert torch.equal(params['gaussian']['logvar'], gp['logvar'])
    assert torch.equal(params['gaussian']['mu_mean'], gp['mu_mean'])
    assert torch.equal(params['gaussian']['logvar_mean'], gp['logvar_mean'])
    kl = O.gauss_kl(params)
    assert torch.equal(kl, g['kl'])
    if 'kl_prior' in g:
        pl = g['prior_logits'].clone().requires_grad_(True)
        _, pparams = O.gauss_forward(pl, g['noise'], training=training)
        klp = O.gauss_kl(params, pparams)
        assert torch.equal(klp, g['kl_prior'])
        (gl,) = torch.autograd.grad((z * g['w']).sum() + kl.sum(), logits, retain_graph=True)
        close(gl, g['grad_logits'])
        gq, gpp = torch.autograd.grad(klp.sum(), [logits, pl])
        close(gq, g['grad_klp_logits'])
        close(gpp, g['grad_klp_prior_logits'])


def test_gaussian_4d(golden):
    """4-D logits: the channel split (isotropic_gaussian.py:108-114), kl over the last dimension only."""
    g = golden('gauss')['gauss_4d']
    logits = g['logits'].clone().requires_grad_(True)
    z, params = O.gauss_forward(logits, g['noise'])
    assert torch.equal(z, g['z']) and z.shape == (3, 3, 5, 8)
    assert torch.equal(params['gaussian']['logvar'], g['params']['gaussian']['logvar'])
    assert torch.equal(params['gaussian']['mu_mean'], g['params']['gaussian']['mu_mean'])
    kl = O.gauss_kl(params)
    assert kl.shape == (3, 3, 5) and torch.equal(kl, g['kl'])
    (gl,) = torch.autograd.grad((z * g['w']).sum() + (kl * g['wk']).sum(), logits)
    close(gl, g['grad_logits'])


def test_gaussian_closed_form(golden):
    """z = e*(0.5*s)+mu (no exp); KL = sum 0.5*(s^2 + mu^2 - 1 - log s^2), finite for s < 0."""
    g = golden('gauss')['gauss_L32_train']
    L = 32
    mu, s = g['logits'][:, :L], g['logits'][:, L:] + g['eps_tol']
    close(g['z'], g['noise'] * (0.5 * s) + mu, rtol=0, atol=0)
    assert (s < 0).any()
    close(g['kl'], (0.5 * (s * s + mu * mu - 1 - torch.log(s * s))).sum(-1), rtol=1e-5, atol=1e-5)


GUMBEL = ['gumbel_K%d_%s_tau%g' % (K, m, t) for K in (5, 32) for m in ('train', 'eval') for t in (1.0, 0.5)]


@pytest.mark.parametrize('name', GUMBEL + ['gumbel_ties'])
def test_gumbel(golden, name):
    g = golden('gumbel')[name]
    logits = g['logits'].clone().requires_grad_(True)
    z, params = O.gumbel_forward(logits, g['U'], tau=g['tau'], training=g['training'])
    assert torch.equal(z, g['z'])
    d, gd = params['discrete'], g['params']['discrete']
    assert torch.equal(d['z_hard'], gd['z_hard'])
    assert torch.equal(d['log_q_z'], gd['log_q_z'])
    assert torch.equal(params['z'], g['params']['z'])
    assert torch.equal((d['z_hard'] == 1).sum(-1), g['hard_count'])
    if name == 'gumbel_ties':
        assert int(g['hard_count'].max()) >= 2  # multi-hot on ties
        return
    assert torch.equal(d['z_hard'].argmax(-1), g['hard_index'])
    kl0 = O.categorical_kl(params)
    assert torch.equal(kl0, g['kl_none'])
    assert float(kl0.detach().abs().max()) == 0.0  # KL(q||q)
    pl = g['prior_logits'].clone().requires_grad_(True)
    prior = {'discrete': {'log_q_z': torch.log_softmax(pl.clone(), -1)}}
    assert torch.equal(prior['discrete']['log_q_z'], g['prior_log_q_z'])
    klp = O.categorical_kl(params, prior)
    assert torch.equal(klp, g['kl_prior'])
    gl, gp = torch.autograd.grad((z * g['w']).sum() + klp.sum(), [logits, pl])
    close(gl, g['grad_logits'])
    close(gp, g['grad_prior_logits'])


def test_gumbel_anneal(golden):
    g = golden('gumbel')['gumbel_anneal']
    tau, it, taus = 1.0, g['start_iteration'], []
    for _ in range(25):
        tau = O.anneal_tau(tau, it, True)
        it += 1
        taus.append(tau)
    assert it == g['end_iteration']
    assert taus == pytest.approx(g['taus'], rel=0, abs=0)


BERN = ['bern_K%d_%s_tau%g' % (K, m, t) for K in (5, 32) for m in ('train', 'eval') for t in (1.0, 0.3)]


@pytest.mark.parametrize('name', BERN)
def test_bernoulli(golden, name):
    g = golden('bern')[name]
    logits = g['logits'].clone().requires_grad_(True)
    z, params = O.bernoulli_forward(logits, g['U'], tau=g['tau'], training=g['training'])
    close(z, g['z'], rtol=1e-6, atol=1e-7)
    assert torch.equal(params['discrete']['z_hard'].detach().round(), g['params']['discrete']['z_hard'].round())
    close(params['z'], g['params']['z'])
    kl0 = O.bernoulli_kl(params)
    close(kl0, g['kl_none'])
    pl = g['prior_logits'].clone().requires_grad_(True)
    klp = O.bernoulli_kl(params, {'discrete': {'logits': pl}})
    close(klp, g['kl_prior'])
    (gl,) = torch.autograd.grad((z * g['w']).sum() + kl0.sum(), logits, retain_graph=True)
    close(gl, g['grad_logits'], rtol=1e-5, atol=1e-6)
    gq, gp = torch.autograd.grad(klp.sum(), [logits, pl])
    close(gq, g['grad_klp_logits'], rtol=1e-5, atol=1e-6)
    close(gp, g['grad_klp_prior_logits'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', ['beta_L4', 'beta_L32'])
def test_beta(golden, name):
    g = golden('beta')[name]
    logits = g['logits'].clone().requires_grad_(True)
    bp = O.beta_params(logits)
    gb = g['params']['beta']
    for k in ('conc1', 'conc2', 'conc1_mean', 'conc2_mean'):
        assert torch.equal(bp[k], gb[k])
    kl0 = O.beta_kl({'beta': bp})
    assert torch.equal(kl0, g['kl_none'])
    pl = g['prior_logits'].clone().requires_grad_(True)
    klp = O.beta_kl({'beta': bp}, {'beta': O.beta_params(pl)})
    assert torch.equal(klp, g['kl_prior'])
    (g0,) = torch.autograd.grad(kl0.sum() + bp['conc1_mean'] + 2 * bp['conc2_mean'], logits, retain_graph=True)
    close(g0, g['grad_kl0_means_logits'])
    gq, gp = torch.autograd.grad(klp.sum(), [logits, pl])
    close(gq, g['grad_klp_logits'])
    close(gp, g['grad_klp_prior_logits'])


@pytest.mark.parametrize('name', ['flowkl_2d', 'flowkl_3d'])
def test_flow_kl(golden, name):
    g = golden('flowkl')[name]
    u = g['u'].clone().requires_grad_(True)
    ld = g['logdet'].clone().requires_grad_(True)
    kl = O.flow_kl({'recon_x': u, 'logdet': ld})
    assert torch.equal(kl, g['kl'])
    gu, gl = torch.autograd.grad((kl * g['w']).sum(), [u, ld])
    close(gu, g['grad_u'])
    close(gl, g['grad_logdet'])
    if name == 'flowkl_2d':
        # the cross-batch mean
        lp = (-0.5 * g['u'] ** 2 - 0.9189385332046727).sum(-1)
        close(g['kl'], (-lp).mean() - g['logdet'][:, 0], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('name', ['concat_train', 'concat_eval'])
def test_concat(golden, name):
    g = golden('concat')[name]
    L, K = 4, 5
    logits = g['logits'].clone().requires_grad_(True)
    children = [('isotropic_gaussian', 2 * L, {}), ('discrete', K, {'tau': 1.0})]
    z, params = O.concat_forward(logits, children, [g['noise'], g['U']], training=g['training'])
    assert torch.equal(z, g['z'])
    assert z.shape[-1] == g['output_size'] and logits.shape[-1] == g['input_size']
    kl = O.concat_kl(params, ['isotropic_gaussian', 'discrete'])
    assert torch.equal(kl, g['kl'])
    for mine, ref in zip(params['concat'], g['params']['concat']):
        assert torch.equal(mine['logits'], ref['logits'])
    assert torch.equal(params['concat'][1]['discrete']['z_hard'], g['params']['concat'][1]['discrete']['z_hard'])
    (gl,) = torch.autograd.grad((z * g['w']).sum() + kl.sum(), logits)
    close(gl, g['grad_logits'])


@pytest.mark.parametrize('name', ['elbo_beta1_plain', 'elbo_beta0.25_extra'])
def test_elbo_combine(golden, name):
    g = golden('elbo')[name]
    res = O.loss_function(g['nll'], g['kld'], g['kl_beta'], extra=g['extra'])
    for k, v in g['result'].items():
        if torch.is_tensor(v):
            assert torch.equal(res[k], v), k
        else:
            assert res[k] == v


def test_vrnn(golden):
    g = golden('vrnn')['vrnn']
    for s in g['steps']:
        z, params = O.vrnn_reparameterize(s['enc_logits'], s['prior_logits'], s['noise_post'], s['noise_prior'])
        assert torch.equal(z['posterior'], s['z_post'])
        assert torch.equal(z['prior'], s['z_prior'])
        assert torch.equal(params['prior']['gaussian']['logvar'], s['params']['prior']['gaussian']['logvar'])
        assert torch.equal(O.vrnn_kld(params, False), s['kld'])
        assert torch.equal(O.vrnn_kld(params, True), s['kld_with_prior_kl'])
    agg = O.vrnn_loss(g['losses'])
    for k, v in g['mean_map'].items():
        if torch.is_tensor(v):
            close(agg[k], v, rtol=1e-6, atol=0)
        else:
            assert agg[k] == pytest.approx(v)


@pytest.mark.parametrize('name', ['vrnnloss_vrnn', 'vrnnloss_additive'])
def test_vrnn_loss_function(golden, name):
    """VRNN.loss_function itself with use_prior_kl, a clamped mutual-information term and a stateful KL annealer
    (vrnn.py:751-765,805-829; abstract_vae.py:319-330,360-363): the oracle's per-step loop against the reference's."""
    g = golden('vrnnloss')[name]
    c = g['config']
    leaves, recons, xs, params = [], [], [], []
    for s in g['steps']:
        enc, pri, xhat, recon = (s[k].clone().requires_grad_(True) for k in ('enc_logits', 'prior_logits',
                                                                             'xhat_logits', 'recon'))
        _, p = O.vrnn_reparameterize(enc, pri, s['noise_post'], s['noise_prior'])
        _, q = O.gauss_forward(xhat, s['noise_xhat'])
        p['posterior']['q_z_given_xhat'] = q
        close(O.vrnn_mut_info(p, c['continuous_mut_info'], c['mut_clamp_strategy'], c['mut_clamp_value']), s['mut_info'])
        close(O.vrnn_kld(p, c['use_prior_kl']), s['kld'], rtol=1e-6, atol=1e-5)
        leaves += [enc, pri, xhat, recon]
        recons.append(recon)
        xs.append(s['x'])
        params.append(p)
    additive = name.endswith('additive')
    out = O.vrnn_loss_function(recons, xs[0] if additive else xs, params, c['nll_type'], g['kl_betas'],
                               use_prior_kl=c['use_prior_kl'], mi_scale=c['continuous_mut_info'],
                               strategy=c['mut_clamp_strategy'], value=c['mut_clamp_value'], additive=additive)
    ref = g['result']
    assert out['count'] == ref['count'] and abs(out['kl_beta_scalar'] - ref['kl_beta_scalar']) < 1e-12
    for k in ('loss', 'elbo', 'loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'additional_loss_mean', 'mut_info_mean'):
        close(out[k], ref[k], rtol=1e-6, atol=1e-5)
    grads = torch.autograd.grad(out['loss_mean'], leaves, allow_unused=True)
    for i, s in enumerate(g['steps']):
        for j, k in enumerate(('enc_logits', 'prior_logits', 'xhat_logits', 'recon')):
            if s['grad_' + k] is None:
                assert grads[4 * i + j] is None or float(grads[4 * i + j].abs().max()) == 0.0
            else:
                close(grads[4 * i + j], s['grad_' + k], rtol=1e-5, atol=1e-7)


def test_multi_loss_k3(golden):
    """abstract_vae.py:347-354 through the reference (stub model): draw order, K-average, combine."""
    g = golden('multi')['multi_lossK3']
    logits = g['logits'].clone().requires_grad_(True)
    W, b = g['dec_weight'].clone().requires_grad_(True), g['dec_bias'].clone().requires_grad_(True)
    z_list, p_list = O.gauss_forward_multi(logits, g['noises'])
    recons = [torch.nn.functional.linear(z, W, b) for z in z_list]
    nll = O.nll_multi(g['x'], recons, 'bernoulli')
    res = O.loss_function(nll, O.gauss_kl(p_list[0]), 1.0)
    for k, v in g['result'].items():
        if torch.is_tensor(v):
            close(res[k], v)
        else:
            assert res[k] == v
    gl, gw, gb = torch.autograd.grad(res['loss_mean'], [logits, W, b])
    close(gl, g['grad_logits'])
    close(gw, g['grad_dec_weight'])
    close(gb, g['grad_dec_bias'])


def test_multi_msg(golden):
    g = golden('multi')['multi_msg']
    z_list, p_list = O.gauss_forward_multi(g['logits'], g['noises'])
    for z, zr in zip(z_list, g['z_list']):
        assert torch.equal(z, zr)
    assert torch.equal(O.gauss_kl(p_list[-1]), g['kl'])
    assert torch.equal(p_list[-1]['gaussian']['logvar'], g['params_last']['gaussian']['logvar'])


@pytest.mark.parametrize('name', ['multi_likelihood_draws', 'multi_likelihood_mean'])
def test_multi_likelihood(golden, name):
    """abstract_vae.py:281-317: expand-one-sample batches, K // B passes, logsumexp - log(K // B)."""
    g = golden('multi')[name]
    x = g['x']
    B = x.shape[0]
    lin = torch.nn.functional.linear
    z_logits = lin(x, g['enc_weight'], g['enc_bias'])
    noises = iter(g['noises'] or [])
    vals = []
    for idx in range(B):
        li = z_logits[idx].expand_as(z_logits).contiguous()
        xi = x[idx].expand_as(x).contiguous()
        elbos = []
        for count in range(g['K'] // B):
            z, p = O.gauss_forward(li, next(noises) if g['reparam_training'] else None, training=g['reparam_training'])
            nll = O.nll(xi, lin(z, g['dec_weight'], g['dec_bias']), 'bernoulli')
            elbos.append(O.loss_function(nll, O.gauss_kl(p), 1.0)['elbo'])
        vals.append(O.iwae_estimate(elbos, count))
    close(torch.mean(torch.stack(vals)), g['value'])


@pytest.mark.parametrize('name', ['gauss', 'gumbel', 'beta'])
def test_mutual_info(golden, name):
    g = golden('mi')['mi_' + name]
    la, lb = g['logits'].clone().requires_grad_(True), g['logits_xhat'].clone().requires_grad_(True)
    if name == 'gauss':
        pa, pb = O.gauss_forward(la, None, training=False)[1], O.gauss_forward(lb, None, training=False)[1]
        mi = O.gauss_mutual_info({**pa, 'q_z_given_xhat': pb}, g['scale'])
    elif name == 'gumbel':
        U = torch.rand_like(la)
        pa, pb = O.gumbel_forward(la, U)[1], O.gumbel_forward(lb, U)[1]
        mi = O.gumbel_mutual_info({**pa, 'q_z_given_xhat': pb}, g['scale'])
    else:
        mi = O.beta_mutual_info({'beta': O.beta_params(la), 'q_z_given_xhat': {'beta': O.beta_params(lb)}}, g['scale'])
    close(mi, g['mi'])
    ga, gb = torch.autograd.grad((mi * g['w']).sum(), [la, lb])
    close(ga, g['grad_logits'], rtol=1e-5, atol=1e-6)
    close(gb, g['grad_logits_xhat'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', ['cond_train_nopkl', 'cond_train_pkl', 'cond_eval_nopkl', 'cond_eval_pkl'])
def test_conditioned_prior(golden, name):
    g = golden('cond')[name]
    logits = g['logits'].clone().requires_grad_(True)
    W, b = g['mlp_weight'].clone().requires_grad_(True), g['mlp_bias'].clone().requires_grad_(True)
    z, params = O.conditioned_reparameterize(logits, g['labels'], W, b, g['num_classes'], g['noise_post'],
                                             g['noise_prior'], training=g['training'])
    assert torch.equal(z, g['z'])
    assert torch.equal(params['prior']['gaussian']['logvar'], g['params']['prior']['gaussian']['logvar'])
    kl = O.vrnn_kld(params, use_prior_kl=g['use_prior_kl'])      # conditioned.py:137-147 is vrnn.py:723-733
    assert torch.equal(kl, g['kld'])
    gl, gw, gb = torch.autograd.grad((z * g['w']).sum() + kl.sum(), [logits, W, b])
    close(gl, g['grad_logits'])
    close(gw, g['grad_mlp_weight'], rtol=1e-5, atol=1e-6)
    close(gb, g['grad_mlp_bias'], rtol=1e-5, atol=1e-6)


# ---- flow-layer tails (flows/__init__.py), tail inputs recorded from the reference's own sub-networks ----
def test_flowtail_made(golden):
    g = golden('flowtail')['flowtail_made']
    Fd = g['x'].shape[1]
    u, ld = O.made_tail(g['x'], g['ma'][:, :Fd], g['ma'][:, Fd:])
    assert torch.equal(u, g['u']) and torch.equal(ld, g['logdet'])


@pytest.mark.parametrize('name', ['flowtail_madesplit_plain', 'flowtail_madesplit_tanh'])
def test_flowtail_madesplit(golden, name):
    g = golden('flowtail')[name]
    u, ld = O.made_tail(g['x'], g['m'], g['a'], pre_exp_tanh=g['pre_exp_tanh'])
    assert torch.equal(u, g['u']) and torch.equal(ld, g['logdet'])


@pytest.mark.parametrize('name', ['flowtail_coupling', 'flowtail_additive'])
def test_flowtail_coupling(golden, name):
    g = golden('flowtail')[name]
    y, ld = O.coupling_tail(g['x'], g.get('s_raw'), g['t_raw'], g['mask'])
    assert torch.equal(y, g['y']) and torch.equal(ld, g['logdet'])


@pytest.mark.parametrize('name', ['flowtail_bn_train', 'flowtail_bn_eval'])
def test_flowtail_bn(golden, name):
    g = golden('flowtail')[name]
    sd = g['state_dict_before']
    x = g['x'].clone().requires_grad_(True)
    lg, beta = sd['log_gamma'].clone().requires_grad_(True), sd['beta'].clone().requires_grad_(True)
    y, ld, mean, var = O.bnflow(x, lg, beta, sd['running_mean'], sd['running_var'], training=g['training'],
                                eps=g['eps'])
    assert torch.equal(y, g['y']) and torch.equal(ld, g['logdet'])
    gx, glg, gb = torch.autograd.grad((y * g['wu']).sum() + (ld.expand(x.shape[0], 1) * g['wl']).sum(), [x, lg, beta])
    close(gx, g['grad_x'])
    close(glg, g['grad_log_gamma'])
    close(gb, g['grad_beta'])
    if g['training']:   # running statistics: flows/__init__.py:240-246
        mom = g['momentum']
        close(sd['running_mean'] * mom + mean.detach() * (1 - mom), g['state_dict_after']['running_mean'])
        close(sd['running_var'] * mom + var.detach() * (1 - mom), g['state_dict_after']['running_var'])
