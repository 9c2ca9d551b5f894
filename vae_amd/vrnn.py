"""The VRNN hot path with the reference's semantics (vrnn.py:395-447 reparameterize /
_clamp_variance, :723-733 kld, :767-829 loss aggregation; additive_vrnn.py:78-104).

The recurrence (LSTM, phi_x / phi_z / prior nets, decoder) is the host project's stock PyTorch;
what is here is what runs per time step on its outputs: two reparameterizations (posterior first,
then the variance-clamped prior), KL(posterior || prior) [+ KL(prior || N(0,1))], and the per-step
ELBO dicts averaged over T.  `loss_function` scores all T steps in ONE fused NLL+ELBO launch over
[T*B] rows instead of T Python-loop iterations of ~40 launches each; the result
equals the reference's `_add_loss_map` / `_mean_map` because every step holds the same B samples.
Every per-step term of the reference's loop is kept: the KL weight of each step (compute_kl_beta runs
once per step: an annealer advances T times per call), the mutual-information term of each step
(vrnn.py:751-765) and an overridden `nll` (then the per-step loop itself runs).  `VRNNMixin` /
`AdditiveVRNNMixin` are what a host class inherits; the free functions serve callers that hold
tensors only.
"""
import torch

from . import functional as F_
from . import rng
from .elbo import ELBOMixin, _pack, clamp_mut_info, vrnn_loss_function
from .lazy import LazyParams
from .reparameterizers.isotropic_gaussian import IsotropicGaussian


def clamp_variance(logits, config, reparameterizer=None):
    """vrnn.py:395-417."""
    if config['reparam_type'] == 'isotropic_gaussian':
        feat_size = logits.size(-1)
        return torch.cat([logits[:, 0:feat_size // 2], torch.sigmoid(logits[:, feat_size // 2:])], -1)
    elif config['reparam_type'] == 'mixture':
        feat_size = reparameterizer.num_continuous_input
        return torch.cat([logits[:, 0:feat_size // 2], torch.sigmoid(logits[:, feat_size // 2:feat_size]),
                          logits[:, feat_size:]], -1)
    return logits


def reparameterize(reparameterizer, logits_map, config, noise=None):
    """vrnn.py:419-447.  `noise` = (posterior noise, prior noise) for injection, else in-kernel Philox;
    both draws consume the generator in the reference's order (posterior first).

    With an IsotropicGaussian reparameterizer on 2-D logits the whole step -- both draws, the variance
    clamp and both KLs -- is ONE fused launch (`functional.gauss_pair`); the KLs ride on the returned
    parameter dicts so that `kld()` below costs nothing, and the two logvar tensors are produced on first read.  Anything else takes the generic two-call path."""
    noise = (None, None) if noise is None else noise
    enc, pri = logits_map['encoder_logits'], logits_map['prior_logits']
    if isinstance(reparameterizer, IsotropicGaussian) and config['reparam_type'] == 'isotropic_gaussian' \
            and enc.dim() == 2 and pri.dim() == 2:
        return _reparameterize_fused(reparameterizer, logits_map, config, noise)
    kw = {} if noise[0] is None else {'noise': noise[0]}
    z_enc_t, params_enc_t = reparameterizer(enc, **kw)
    logits_map['prior_logits'] = clamp_variance(pri, config, reparameterizer)
    kw = {} if noise[1] is None else {'noise': noise[1]}
    z_prior_t, params_prior_t = reparameterizer(logits_map['prior_logits'], **kw)
    z = {'prior': z_prior_t, 'posterior': z_enc_t, 'x_features': logits_map['x_features']}
    params = {'prior': params_prior_t, 'posterior': params_enc_t}
    return z, params


def _reparameterize_fused(rep, logits_map, config, noise, force=False):
    enc, pri = logits_map['encoder_logits'], logits_map['prior_logits']
    L = rep.output_size
    assert enc.shape[-1] == 2 * L and pri.shape[-1] == 2 * L
    training = bool(rep.training or force)
    seed, offset = (0, 0)
    if training and (noise[0] is None or noise[1] is None):
        seed, offset = rng.next_call()
    z_q, z_p, _, _, clamped, kl_pair, kl_prior, means = F_.gauss_pair(
        enc, pri, noise[0], noise[1], eps_tol=rep.eps_tol, training=training, with_prior_kl=True, seed=seed,
        offset=offset, row0=rep.row_offset, mean_scale=rep.mean_scale, want_logvars=False,
        want_means=bool(config.get('vae_amd_step_means', False)))
    mu_q = torch.narrow(enc, -1, 0, L)
    mu_p = torch.narrow(clamped, -1, 0, L)
    if not training:
        z_q, z_p = mu_q, mu_p
    g_q = LazyParams({'z': z_q, 'mu': mu_q})
    g_p = LazyParams({'z': z_p, 'mu': mu_p})
    # the logvar entries are the reference's own ops (isotropic_gaussian.py:124-125 on the encoder logits and on the
    # clamped prior logits), run only if somebody reads them: both KLs already came out of the launch
    eps = rep.eps_tol
    g_q.set_lazy('logvar', lambda: torch.narrow(enc, -1, L, L) + eps)
    g_p.set_lazy('logvar', lambda: torch.narrow(clamped, -1, L, L) + eps)
    if means is not None:
        means = means.to(enc.dtype)
        g_q['mu_mean'], g_q['logvar_mean'], g_p['mu_mean'], g_p['logvar_mean'] = means[0], means[1], means[2], means[3]
    else:
        # the four diagnostics of isotropic_gaussian.py:213-214, by the reference's own op, if and when they are read:
        # a per-step launch of a few thousand rows is latency-bound and the grid-wide sum's closing chain was a third
        # of it (config['vae_amd_step_means'] = True puts them back into the launch)
        ms = rep.mean_scale
        red = (lambda t: torch.mean(t)) if ms is None else (lambda t, c=ms: torch.sum(t) * c)
        g_q.set_lazy('mu_mean', lambda: red(mu_q))
        g_q.set_lazy('logvar_mean', lambda: red(g_q['logvar']))
        g_p.set_lazy('mu_mean', lambda: red(mu_p))
        g_p.set_lazy('logvar_mean', lambda: red(g_p['logvar']))
    g_q._vae_amd_kl_pair = (kl_pair, g_p)        # KL(post || prior), valid against exactly this prior
    g_p._vae_amd_kl = (kl_prior, mu_p)           # KL(prior || N(0,1))
    logits_map['prior_logits'] = clamped
    params_q = {'z': z_q, 'logits': enc, 'gaussian': g_q}
    params_p = {'z': z_p, 'logits': clamped, 'gaussian': g_p}
    z = {'prior': z_p, 'posterior': z_q, 'x_features': logits_map['x_features']}
    return z, {'prior': params_p, 'posterior': params_q}


def kld(reparameterizer, dist, config):
    """vrnn.py:723-733."""
    prior_kl = reparameterizer.kl(dist['prior']) if config['use_prior_kl'] is True else 0
    return reparameterizer.kl(dist['posterior'], dist['prior']) + prior_kl


loss_function_aggregate = vrnn_loss_function   # vrnn.py:767-803 on already computed per-step dicts (one implementation: vae_amd.elbo)


def mut_info(reparameterizer, config, dist_params):
    """vrnn.py:751-765: the clamped mutual-information term of one step from its posterior's parameters (which carry
    'q_z_given_xhat', vrnn.py:735-749), or None -- the reference's zeros[B] -- while both weights are 0."""
    if config.get('continuous_mut_info', 0) > 0 or config.get('discrete_mut_info', 0) > 0:
        return clamp_mut_info(config, reparameterizer.mutual_info(dist_params['posterior']))
    return None


def loss_function(reparameterizer, config, recon_x, x, params, K=1, kl_beta=None, additive=False, inv_count=None,
                  model=None):
    """vrnn.py:805-829 (additive=True: additive_vrnn.py:78-104, every step scored on the LAST
    reconstruction).  recon_x / params: per-step lists; x: per-step list or one tensor for all steps.

    The reference runs AbstractVAE.loss_function once per step (vrnn.py:825-827), i.e. per step: nll, self.kld,
    self.mut_info (abstract_vae.py:360 -> vrnn.py:751-765), self.compute_kl_beta (abstract_vae.py:363: a stateful
    annealer advances once per STEP and every step is weighted by its own value).  All of it is kept: `model` (a
    VRNNMixin-style host object) supplies kld / mut_info / compute_kl_beta and they are called once per step, in step
    order; without a model the same terms come from `config` (`kl_beta`: one float or T floats overrides
    config['kl_beta']).  The T steps are then scored by ONE launch pair over [T*B] rows that reads the T tensors where
    they are.  A model that overrides `nll`, or T > 128 steps, takes the reference's own per-step loop."""
    assert len(recon_x) == len(params)
    assert K == 1, "Monte carlo sampling for decoding not implemented for VRNN"
    T = len(recon_x)
    if not isinstance(x, list) and len(x) != len(recon_x):
        x = [x for _ in range(T)]
        if additive:
            recon_x = [recon_x[-1] for _ in range(T)]
    if model is not None and (T > F_.MAX_STEPS or model._own_nll()):
        return vrnn_loss_function([ELBOMixin.loss_function(model, r, xi, p, K=K) for r, xi, p in zip(recon_x, x, params)])
    B = recon_x[0].shape[0]
    klds, mis, betas = [], [], []
    for t, p in enumerate(params):
        klds.append(model.kld(p) if model is not None else kld(reparameterizer, p, config))
        mis.append(model.mut_info(p, B) if model is not None else mut_info(reparameterizer, config, p))
        if model is not None:
            betas.append(model.compute_kl_beta([config['kl_beta']])[0])
        elif isinstance(kl_beta, (list, tuple)):
            betas.append(kl_beta[t])
        else:
            betas.append(config['kl_beta'] if kl_beta is None else kl_beta)
    klds = torch.cat(klds, 0)                                                           # [T*B]: 4 bytes per step and sample
    if all(m is None for m in mis):
        mis = None
    else:
        mis = torch.cat([torch.zeros_like(klds[:B]) if m is None else m for m in mis], 0)
    betas = [float(b) for b in betas]
    uniform = all(b == betas[0] for b in betas)
    # 'kl_beta_scalar' of the aggregate: the per-step values summed one after the other, then / T (vrnn.py:781-801)
    beta_scalar = betas[0]
    for b in betas[1:]:
        beta_scalar = beta_scalar + b
    beta_scalar = beta_scalar / T
    if inv_count is None:
        inv_count = 1.0 / (T * B)
    else:
        inv_count = inv_count / T
    if T <= F_.MAX_STEPS:
        # the T reconstructions and targets are read where the decoder / the loader left them (pointer tables in the
        # kernel arguments); the per-sample averages over the steps come out of the closing launch
        _, _, _, means, loss_avg, elbo_avg = F_.nll_steps_elbo(list(x), list(recon_x), config['nll_type'], kld=klds,
                                                               mi=mis, kl_beta=betas[0] if uniform else betas,
                                                               inv_count=inv_count)
        out = _pack(loss_avg, elbo_avg, means, beta_scalar, loss_avg.dtype)
    else:
        if not uniform:
            raise NotImplementedError("vae_amd.vrnn.loss_function: more than %d steps with a per-step KL weight needs "
                                      "a host model (model=...) for the per-step loop" % F_.MAX_STEPS)
        recon_all = torch.cat([r.reshape(B, -1) for r in recon_x], 0)                  # [T*B, D or 2D]
        x_all = torch.cat([xi.reshape(B, -1) for xi in x], 0)
        _, loss, elbo, means = F_.nll_elbo(x_all, recon_all, config['nll_type'], kld=klds, mi=mis, kl_beta=betas[0],
                                           inv_count=inv_count)
        out = _pack(loss.view(T, B).mean(0), elbo.view(T, B).mean(0), means, beta_scalar, loss.dtype)
    out['count'] = T
    return out


class VRNNMixin(ELBOMixin):
    """Drop-in for the per-step half of VRNN (vrnn.py:395-447 reparameterize, :723-733 kld, :751-765 mut_info,
    :805-829 loss_function); the recurrence, the networks and generation stay the host's.  A host class lists it
    first: `class VRNN(VRNNMixin, AbstractVAE)`."""

    additive = False   # AdditiveVRNN (additive_vrnn.py:78-104): every step scored on the last reconstruction

    def _clamp_variance(self, logits):
        return clamp_variance(logits, self.config, self.reparameterizer)

    def reparameterize(self, logits_map, noise=None):
        return reparameterize(self.reparameterizer, logits_map, self.config, noise=noise)

    def kld(self, dist):
        return kld(self.reparameterizer, dist, self.config)

    def mut_info(self, dist_params, batch_size):
        return mut_info(self.reparameterizer, self.config, dist_params)

    def loss_function(self, recon_x, x, params, K=1):
        out = loss_function(self.reparameterizer, self.config, recon_x, x, params, K=K, additive=self.additive,
                            inv_count=self.inv_count, model=self)
        self.nan_check()
        return out


class AdditiveVRNNMixin(VRNNMixin):
    additive = True
