"""CPU oracle: an op-for-op PyTorch restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under vae_amd/ imports this file; only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg do, and only as the checker /
the timed CPU baseline, never as the product path.

Every function restates, unfused and in the reference's op order, one piece of
/root/reference (cited per function), with the random draw turned into an explicit
argument so the HIP kernels can be fed identical noise.  Pinning: tests/golden/*.pt hold
outputs of the reference itself (imported in the build container by
tests/golden/make_golden.py); tests/test_oracle_golden.py checks this file against them.

NOT pinned by anything in the reference ("parity unpinned"):
  * the three NLLs (`nll_bernoulli`, `nll_gaussian`, `nll_disc_logistic`): the reference
    delegates to `helpers.distributions.nll` (abstract_vae.py:471), a second submodule
    (github.com/jramapuram/helpers, no pinned version) that is absent from /root/reference.
    The definitions below are this build's stated contract, written from torch primitives.
  * `EPS_TOL` = helpers.utils.eps(half) (isotropic_gaussian.py:125): assumed 1e-6 for fp32.
"""
import math

import torch
import torch.nn.functional as F
import torch.distributions as D

EPS_TOL = 1e-6

# the reference's KL needs validation off (scale <= 0 is legal there)
D.Distribution.set_default_validate_args(False)


# ----------------------------------------------------------------------------- gaussian
def _slice_dim(logits):
    # isotropic_gaussian.py:108-114, beta.py:90-96, concat_reparameterizer.py:102-108
    return {2: -1, 3: -1, 4: 1}[logits.dim()]


def gauss_forward(logits, noise, training=True, eps_tol=EPS_TOL):
    """isotropic_gaussian.py:98-131 (split, + eps), :77-96 (z = eps*(0.5*s) + mu, NO exp), :204-215."""
    dim = _slice_dim(logits)
    feat = logits.shape[dim]
    assert feat % 2 == 0
    mu = torch.narrow(logits, dim, 0, feat // 2)
    sigma = torch.narrow(logits, dim, feat // 2, feat // 2)
    sigma = sigma + eps_tol
    if training:
        std = sigma.mul(0.5)
        eps = noise.type(std.dtype)
        z = eps.mul(std).add_(mu)
    else:
        z = mu
    g = {'z': z, 'mu': mu, 'logvar': sigma}
    g['mu_mean'] = torch.mean(mu)
    g['logvar_mean'] = torch.mean(sigma)
    return z, {'z': z, 'logits': logits, 'gaussian': g}


def gauss_kl(dist_a, prior=None):
    """isotropic_gaussian.py:158-190 via torch/distributions/kl.py:468-471."""
    mu, s = dist_a['gaussian']['mu'], dist_a['gaussian']['logvar']
    if prior is None:
        p = D.Normal(torch.zeros_like(mu), torch.ones_like(s))
    else:
        p = D.Normal(prior['gaussian']['mu'], prior['gaussian']['logvar'])
    return torch.sum(D.kl_divergence(D.Normal(mu, s), p).type(mu.dtype), -1)


# ------------------------------------------------------------------------------- gumbel
def anneal_tau(tau, iteration, training, tau0=1.0, anneal_rate=3e-6, min_temp=0.5, interval=10):
    """gumbel.py:91-108 / bernoulli.py:80-97 (min_temp 0.5 / 1e-6)."""
    if training and iteration > 0 and iteration % interval == 0:
        return float(max(tau0 * math.exp(-anneal_rate * iteration), min_temp))
    return tau


def gumbel_forward(logits, U, tau=1.0, training=True, eps=1e-9):
    """gumbel.py:110-124, :257-304, :317-341.  U is the torch.rand draw of :269."""
    log_q_z = F.log_softmax(logits.clone(), dim=-1)
    noise = U.clone().float() if U.dtype != logits.dtype else U.clone()
    noise.add_(eps).log_().neg_()
    noise.add_(eps).log_().neg_()
    x = (logits + noise.type(logits.dtype)) / tau
    y = F.softmax(x + eps, dim=-1)
    y_max, _ = torch.max(y, dim=-1, keepdim=True)
    y_hard = torch.eq(y_max.data, y.data).type(logits.dtype)
    y_hard = (y_hard - y).detach() + y
    params = {'z_hard': y_hard, 'logits': logits, 'log_q_z': log_q_z, 'tau_scalar': tau}
    if training:
        return y, {'z': y, 'discrete': params}
    return y_hard, {'z': y, 'discrete': params}


def categorical_kl(dist_a, prior=None):
    """gumbel.py:222-254: sum softmax(a) * (log_softmax(a) - log_softmax(b)), a/b re-normalised by
    OneHotCategorical(logits=...) (torch/distributions/categorical.py: logits - logsumexp)."""
    a = D.OneHotCategorical(logits=dist_a['discrete']['log_q_z'])
    b = a if prior is None else D.OneHotCategorical(logits=prior['discrete']['log_q_z'])
    if prior is None:
        b = D.OneHotCategorical(logits=dist_a['discrete']['log_q_z'])
    return torch.sum(F.softmax(a.logits, dim=-1) * (F.log_softmax(a.logits, dim=-1)
                                                     - F.log_softmax(b.logits, dim=-1)), -1)


# ---------------------------------------------------------------------------- bernoulli
def _clamp_probs(p):
    e = torch.finfo(p.dtype).eps
    return p.clamp(min=e, max=1 - e)


def bernoulli_forward(logits, U, tau=1.0, training=True):
    """bernoulli.py:111-136, :196-212; RelaxedBernoulli.rsample =
    torch/distributions/relaxed_bernoulli.py:104-112 with U the torch.rand draw."""
    probs = _clamp_probs(torch.sigmoid(logits))          # logits_to_probs(..., is_binary=True) + clamp_probs
    u = _clamp_probs(U.type(logits.dtype))
    lg = (u.log() - (-u).log1p() + probs.log() - (-probs).log1p()) / tau
    relaxed = torch.sigmoid(lg)                          # SigmoidTransform of the LogitRelaxedBernoulli sample
    hard = relaxed.clone()
    hard[relaxed < 0.5] = 0.0
    hard[relaxed >= 0.5] = 1.0
    hard = (hard - relaxed).detach() + relaxed
    params = {'z_hard': hard, 'logits': logits, 'tau_scalar': tau}
    z_out = relaxed if training else hard
    return z_out, {'z': relaxed, 'logits': logits, 'discrete': params}


def bernoulli_kl(dist_a, prior=None, eps=1e-6):
    """bernoulli.py:147-191."""
    q1 = torch.sigmoid(dist_a['discrete']['logits'])
    if prior is None:
        p1 = torch.zeros_like(q1) + 0.5
    else:
        p1 = torch.sigmoid(prior['discrete']['logits'])
    q0, p0 = 1 - q1, 1 - p1
    k1 = q1 * ((q1 + eps).log() - (p1 + eps).log())
    k0 = q0 * ((q0 + eps).log() - (p0 + eps).log())
    return torch.sum(k1 + k0, -1)


# --------------------------------------------------------------------------------- beta
def beta_params(logits):
    """beta.py:78-108 (sigmoid of both halves), :166-177 (means).  The sample itself is
    torch._sample_dirichlet (not injectable): callers draw it with D.Beta(conc1, conc2)."""
    dim = _slice_dim(logits)
    feat = logits.shape[dim]
    conc1 = torch.sigmoid(torch.narrow(logits, dim, 0, feat // 2))
    conc2 = torch.sigmoid(torch.narrow(logits, dim, feat // 2, feat // 2))
    return {'conc1': conc1, 'conc2': conc2, 'conc1_mean': torch.mean(conc1), 'conc2_mean': torch.mean(conc2)}


def beta_kl(dist_a, prior=None):
    """beta.py:110-136 via torch/distributions/kl.py:219-228 (_kl_beta_beta)."""
    c1, c0 = dist_a['beta']['conc1'], dist_a['beta']['conc2']
    if prior is None:
        p = D.Beta(torch.zeros_like(c1) + 1.1, torch.zeros_like(c0) + 1.1)
    else:
        p = D.Beta(prior['beta']['conc1'], prior['beta']['conc2'])
    return torch.sum(D.kl_divergence(D.Beta(c1, c0), p), -1)


# --------------------------------------------------------------------------------- flow
def flow_kl(dist_a):
    """flow.py:67-78.  With 2-D logits `logdet` is [B,1] and the subtraction broadcasts to [B,B]
    (SURVEY Q6): kl_i = mean_j(-lp_j) - logdet_i.  With [B,R] logdet it is a plain row mean."""
    recon, logdet = dist_a['recon_x'], dist_a['logdet']
    nll = (-0.5 * recon.pow(2) - 0.5 * math.log(2 * math.pi)).sum(-1)
    return torch.mean(-nll - logdet, -1)


# ------------------------------------------------------------------- concat / sequential
def concat_forward(logits, children, noises, training=True):
    """concat_reparameterizer.py:90-118.  children: list of (kind, input_size, kwargs)."""
    dim = _slice_dim(logits)
    outs, plist, begin = [], [], 0
    for (kind, size, kw), noise in zip(children, noises):
        li = torch.narrow(logits, dim, begin, size)
        zi, pi = LEAF_FORWARD[kind](li, noise, training=training, **kw)
        outs.append(zi.clone())
        plist.append({**pi, 'logits': logits[:, begin:begin + size]})
        begin += size
    return torch.cat(outs, -1), {'logits': logits, 'concat': plist}


def concat_kl(dists, kinds, priors=None):
    """concat_reparameterizer.py:71-88."""
    plist = dists['concat']
    priors = [None] * len(plist) if priors is None else priors
    kl = None
    for p, pr, kind in zip(plist, priors, kinds):
        k = LEAF_KL[kind](p, pr)
        kl = k if kl is None else kl + k
    return kl


LEAF_FORWARD = {
    'isotropic_gaussian': lambda l, n, training=True, **kw: gauss_forward(l, n, training=training, **kw),
    'discrete': lambda l, n, training=True, **kw: gumbel_forward(l, n, training=training, **kw),
    'bernoulli': lambda l, n, training=True, **kw: bernoulli_forward(l, n, training=training, **kw),
}
LEAF_KL = {'isotropic_gaussian': gauss_kl, 'discrete': categorical_kl, 'bernoulli': bernoulli_kl, 'beta': beta_kl}


# ---------------------------------------------------------------------------------- nll
def nll_bernoulli(x, recon_logits):
    """-sum Bernoulli(logits=recon).log_prob(x) over all non-batch dims.  (assumed, see header)"""
    B = x.size(0)
    return -torch.sum(D.Bernoulli(logits=recon_logits.reshape(B, -1)).log_prob(x.reshape(B, -1)), -1)


def _split_variance(x, recon):
    # abstract_vae.py:33-35: decoder output is cat([mean, variance.expand_as(mean)], 1)
    B, half = x.size(0), recon.size(1) // 2
    return recon[:, :half].reshape(B, -1), recon[:, half:].reshape(B, -1)


def nll_gaussian(x, recon):
    """-sum Normal(mean, softplus(v) + 1e-6).log_prob(x).  (assumed, see header)"""
    B = x.size(0)
    m, v = _split_variance(x, recon)
    return -torch.sum(D.Normal(m, F.softplus(v) + 1e-6).log_prob(x.reshape(B, -1)), -1)


def nll_disc_logistic(x, recon):
    """256-bin discretized logistic: -sum log(sigmoid((xq + 1/256 - m)/s) - sigmoid((xq - m)/s) + 1e-7),
    xq = floor(256 x)/256, m = sigmoid(mean logits), s = exp(hardtanh(v, -4.5, 0)).  (assumed, see header)"""
    B = x.size(0)
    a, v = _split_variance(x, recon)
    binsize = 1.0 / 256.0
    mean = torch.sigmoid(a)
    scale = torch.exp(F.hardtanh(v, min_val=-4.5, max_val=0.0))
    xs = (torch.floor(x.reshape(B, -1) / binsize) * binsize - mean) / scale
    cdf_plus = torch.sigmoid(xs + binsize / scale)
    cdf_minus = torch.sigmoid(xs)
    return -torch.sum(torch.log(cdf_plus - cdf_minus + 1e-7), -1)


NLL = {'bernoulli': nll_bernoulli, 'gaussian': nll_gaussian, 'disc_logistic': nll_disc_logistic,
       'log_logistic_256': nll_disc_logistic}


def nll(x, recon_x, nll_type):
    """abstract_vae.py:461-471."""
    return NLL[nll_type.strip().lower()](x.contiguous(), recon_x.contiguous())


# --------------------------------------------------------------------------------- elbo
def loss_function(nll_, kld, kl_beta=1.0, mut_info=None, extra=None):
    """abstract_vae.py:356-388 given the per-sample nll and kld."""
    elbo = nll_ + kld
    mut_info = torch.zeros_like(nll_) if mut_info is None else mut_info
    additional = torch.sum(torch.cat([v.unsqueeze(0) for v in extra.values()], 0), 0) \
        if extra else torch.zeros_like(nll_)
    loss = (nll_ + additional + kl_beta * kld) - mut_info
    return {
        'loss': loss, 'elbo': elbo,
        'loss_mean': torch.mean(loss), 'elbo_mean': torch.mean(elbo), 'nll_mean': torch.mean(nll_),
        'kld_mean': torch.mean(kld), 'additional_loss_mean': torch.mean(additional),
        'kl_beta_scalar': kl_beta, 'mut_info_mean': torch.mean(mut_info),
    }


# --------------------------------------------------------------------------------- vrnn
def gauss_mutual_info(dist, scale):
    """isotropic_gaussian.py:142-156: KL(N(x params) || N(xhat params)) as a cross-entropy proxy."""
    a, b = dist['gaussian'], dist['q_z_given_xhat']['gaussian']
    return scale * torch.sum(D.kl_divergence(D.Normal(a['mu'], a['logvar']), D.Normal(b['mu'], b['logvar'])), -1)


def gumbel_mutual_info(dist, scale):
    """gumbel.py:143-168 (the Monte-Carlo form is the live one): sum p_z (log q(z|xhat) - log q(z|x))."""
    p_z = F.softmax(dist['discrete']['logits'], dim=-1)
    cross = torch.sum(p_z * dist['q_z_given_xhat']['discrete']['log_q_z'], -1)
    ent = -torch.sum(p_z * dist['discrete']['log_q_z'], -1)
    return scale * (ent + cross)


def beta_mutual_info(dist, scale):
    """beta.py:138-152: KL(Beta(xhat params) || Beta(x params))."""
    a, b = dist['beta'], dist['q_z_given_xhat']['beta']
    return scale * torch.sum(D.kl_divergence(D.Beta(b['conc1'], b['conc2']), D.Beta(a['conc1'], a['conc2'])), -1)


def nll_multi(x, recon_list, nll_type):
    """abstract_vae.py:344-354: nll of the first decode, += nll of each further draw's decode, / K."""
    out = nll(x, recon_list[0], nll_type)
    for r in recon_list[1:]:
        out = out + nll(x, r, nll_type)
    return out / len(recon_list)


def gauss_forward_multi(logits, noises, eps_tol=EPS_TOL):
    """msg.py:63-69 / abstract_vae.py:348-350: the reparameterizer called once per draw on the same logits."""
    pairs = [gauss_forward(logits, n, training=True, eps_tol=eps_tol) for n in noises]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def iwae_estimate(elbos, count):
    """abstract_vae.py:313-315: logsumexp of the stacked elbo values minus log(count + 1), where `count` is
    the last index of the draw loop (NOT the number of values: each of the K // B passes contributes B)."""
    flat = torch.cat([e.unsqueeze(0) for e in elbos], 0).view([-1])
    return torch.logsumexp(flat, dim=0) - math.log(count + 1)


def clamp_variance(logits):
    """vrnn.py:395-408 (isotropic_gaussian branch)."""
    feat = logits.size(-1)
    return torch.cat([logits[:, 0:feat // 2], torch.sigmoid(logits[:, feat // 2:])], -1)


def vrnn_reparameterize(enc_logits, prior_logits, noise_post, noise_prior, training=True):
    """vrnn.py:419-447: posterior draw first, then the clamped prior."""
    z_enc, p_enc = gauss_forward(enc_logits, noise_post, training=training)
    z_pri, p_pri = gauss_forward(clamp_variance(prior_logits), noise_prior, training=training)
    return {'prior': z_pri, 'posterior': z_enc}, {'prior': p_pri, 'posterior': p_enc}


def conditioned_reparameterize(logits, labels, mlp_weight, mlp_bias, num_classes, noise_post, noise_prior,
                               training=True):
    """conditioned.py:89-135: posterior from the encoder logits; prior from one-hot labels -> dense projector ->
    sigmoid-clamped scale half (:44-57) -> the same reparameterizer.  Posterior is drawn first."""
    z_q, p_q = gauss_forward(logits, noise_post, training=training)
    onehot = F.one_hot(labels, num_classes).type(logits.dtype)
    prior_logits = clamp_variance(F.linear(onehot, mlp_weight, mlp_bias))
    _, p_p = gauss_forward(prior_logits, noise_prior, training=training)
    return z_q, {'labels': labels, 'logits': logits, 'prior': p_p, 'posterior': p_q}


def vrnn_kld(dist, use_prior_kl=False):
    """vrnn.py:723-733."""
    prior_kl = gauss_kl(dist['prior']) if use_prior_kl else 0
    return gauss_kl(dist['posterior'], dist['prior']) + prior_kl


def vrnn_loss(step_losses):
    """vrnn.py:767-829: sum the per-step dicts, divide every entry by the step count."""
    agg = None
    for lt in step_losses:
        if agg is None:
            agg = {**lt, 'count': 1}
        else:
            for k, v in lt.items():
                agg[k] = agg[k] + v
            agg['count'] += 1
    for k in agg:
        if k != 'count':
            agg[k] = agg[k] / agg['count']
    return agg


def clamp_mut_info(mut_info, strategy='none', value=None):
    """abstract_vae.py:473-488."""
    strategy = strategy.strip().lower()
    if strategy == 'norm':
        return mut_info / torch.norm(mut_info, p=2)
    if strategy == 'clamp':
        return torch.clamp(mut_info, min=-value, max=value)
    assert strategy == 'none'
    return mut_info


def vrnn_mut_info(dist_params, scale, strategy='none', value=None):
    """vrnn.py:751-765 (IsotropicGaussian reparameterizer): zeros unless the weight is > 0, else the clamped
    mutual-information proxy of the POSTERIOR's parameters."""
    if scale > 0:
        return clamp_mut_info(gauss_mutual_info(dist_params['posterior'], scale), strategy, value)
    return torch.zeros(dist_params['posterior']['gaussian']['mu'].shape[0])


def vrnn_loss_function(recons, xs, params, nll_type, kl_betas, use_prior_kl=False, mi_scale=0.0, strategy='none',
                       value=None, additive=False, nll_fn=None):
    """vrnn.py:805-829 / additive_vrnn.py:78-104: AbstractVAE.loss_function (abstract_vae.py:332-388) once per step --
    nll, VRNN.kld, VRNN.mut_info, that step's KL weight (compute_kl_beta is called inside the step: `kl_betas` is
    the sequence a stateful annealer produced) -- then _add_loss_map / _mean_map.  xs: a list, or one tensor scored
    at every step (additive: then every step also uses the LAST reconstruction)."""
    T = len(recons)
    if not isinstance(xs, list):
        xs = [xs for _ in range(T)]
        if additive:
            recons = [recons[-1] for _ in range(T)]
    nll_fn = nll_fn or (lambda x_, r_: nll(x_, r_, nll_type))
    steps = []
    for r, x_, p, beta in zip(recons, xs, params, kl_betas):
        steps.append(loss_function(nll_fn(x_, r), vrnn_kld(p, use_prior_kl), kl_beta=beta,
                                   mut_info=vrnn_mut_info(p, mi_scale, strategy, value)))
    return vrnn_loss(steps)


# --------------------------------------------------------------------------- flow tails
def made_tail(x, m, a, pre_exp_tanh=False):
    """flows/__init__.py:121-125 (MADESplit) / :176-178 (MADE), 'direct': -> (u, logdet [B,1])."""
    if pre_exp_tanh:
        a = torch.tanh(a)
    u = (x - m) * torch.exp(-a)
    return u, -a.sum(-1, keepdim=True)


def coupling_tail(x, s_raw, t_raw, mask):
    """flows/__init__.py:524-528 (s_raw None: the additive layer, :458-460): -> (y, logdet [B,1])."""
    t = t_raw * (1 - mask)
    if s_raw is None:
        return x + t, torch.zeros_like(t).sum(-1, keepdim=True)
    log_s = s_raw * (1 - mask)
    return x * torch.exp(log_s) + t, log_s.sum(-1, keepdim=True)


def bnflow(x, log_gamma, beta, running_mean, running_var, training=True, eps=1e-5):
    """flows/__init__.py:233-257 'direct': -> (y, logdet [1], mean, var) (mean/var: what the running
    statistics are updated with in training, :240-246)."""
    if training:
        mean = x.mean(0)
        var = (x - mean).pow(2).mean(0) + eps
    else:
        mean, var = running_mean, running_var
    x_hat = (x - mean) / var.sqrt()
    y = torch.exp(log_gamma) * x_hat + beta
    return y, (log_gamma - 0.5 * torch.log(var)).sum(-1, keepdim=True), mean, var


def reverse_columns(x):
    """flows/__init__.py:403-406."""
    return x.flip(-1)


# ------------------------------------------------------------- the bench's CPU baseline
def elbo_step(logits, noise, x, recon, nll_type='bernoulli', kl_beta=1.0, backward=True):
    """One SimpleVAE ELBO step of the hot path on the CPU, unfused as in the reference:
    gaussian reparameterize -> kl -> nll -> loss dict (-> backward of loss_mean + a dummy decoder use of z)."""
    if backward:
        logits = logits.detach().requires_grad_(True)
        recon = recon.detach().requires_grad_(True)
    z, params = gauss_forward(logits, noise, training=True)
    n = nll(x, recon, nll_type)
    k = gauss_kl(params)
    out = loss_function(n, k, kl_beta)
    if backward:
        # the decoder's gradient w.r.t. z is stood in for by a unit cotangent (as in bench.py)
        (out['loss_mean'] + z.sum() * 0.0 + z.mean()).backward()
        return out, logits.grad, recon.grad
    return out, None, None
