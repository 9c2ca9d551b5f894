"""AbstractVAE's ELBO methods with the reference's signatures (abstract_vae.py:319-388, :400-410,
:451-471, :506-520) and the VRNN aggregation (vrnn.py:723-733, :767-829; additive_vrnn.py:78-104),
routed through the fused NLL+ELBO kernels.

`ELBOMixin` is what a host model class inherits (it supplies loss_function / nll / kld /
reparameterize / mut_info / compute_kl_beta and expects `self.config`, `self.reparameterizer`,
optionally `self.kl_annealer`, `self.decode`).  The free functions are the same code for callers
that hold tensors only.
"""
import torch

from . import functional as F_

MEAN_KEYS = ('loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'additional_loss_mean', 'mut_info_mean')


def nll(x, recon_x, nll_type):
    """[B] negative log-likelihood (abstract_vae.py:461-471)."""
    return F_.nll(x, recon_x, nll_type)


def _pack(loss, elbo, means, kl_beta, dtype):
    means = means.to(dtype)
    return {
        'loss': loss,
        'elbo': elbo,
        'loss_mean': means[0],
        'elbo_mean': means[1],
        'nll_mean': means[2],
        'kld_mean': means[3],
        'additional_loss_mean': means[4],
        'kl_beta_scalar': kl_beta,
        'mut_info_mean': means[5],
    }


def loss_function(recon_x, x, kld, nll_type, kl_beta=1.0, mut_info=None, extra_loss_terms=None, inv_count=None,
                  nll_given=None):
    """The loss dict of abstract_vae.py:377-388 from tensors.  One fused launch (+ a 1-block finalize)
    when the NLL comes from (x, recon_x); `nll_given` [B] (K > 1 average) uses the combine kernel."""
    add = None
    if extra_loss_terms:
        vals = list(extra_loss_terms.values())
        add = vals[0] if len(vals) == 1 else torch.sum(torch.stack(vals, 0), 0)
    if nll_given is None:
        _, loss, elbo, means = F_.nll_elbo(x, recon_x, nll_type, kld=kld, add=add, mi=mut_info, kl_beta=kl_beta,
                                           inv_count=inv_count)
    else:
        loss, elbo, means = F_.elbo_combine(nll_given, kld=kld, add=add, mi=mut_info, kl_beta=kl_beta,
                                            inv_count=inv_count)
    return _pack(loss, elbo, means, kl_beta, loss.dtype)


def vrnn_loss_function(step_losses):
    """vrnn.py:767-829: per-step loss dicts summed, then every entry divided by the step count."""
    agg = None
    for lt in step_losses:
        if agg is None:
            agg = {**lt, 'count': 1}
            continue
        for k, v in lt.items():
            agg[k] = agg[k] + v
        agg['count'] += 1
    for k in agg.keys():
        if k != 'count':
            agg[k] = agg[k] / agg['count']
    return agg


class ELBOMixin(object):
    """Drop-in for the ELBO half of AbstractVAE (the encoder/decoder half stays the host's)."""

    kl_annealer = None
    inv_count = None  # 1/B_global when the batch is sharded (vae_amd.dist)

    def compute_kl_beta(self, kl_beta_list):
        if self.kl_annealer is not None:
            kl_beta_list = self.kl_annealer(kl_beta_list)
        return kl_beta_list

    def reparameterize(self, logits, labels=None, force=False):
        return self.reparameterizer(logits, force=force)

    def kld(self, dist_a):
        return self.reparameterizer.kl(dist_a)

    def nll(self, x, recon_x, nll_type):
        return nll(x, recon_x, nll_type)

    def mut_info(self, dist_params, batch_size):
        if self.config.get('continuous_mut_info', 0) > 0 or self.config.get('discrete_mut_info', 0) > 0:
            return self._clamp_mut_info(self.reparameterizer.mutual_info(dist_params))
        return None  # the reference's zeros[B]; the kernel treats a missing term as 0

    def _clamp_mut_info(self, mut_info):
        strategy = self.config['mut_clamp_strategy'].strip().lower()
        if strategy == 'norm':
            return mut_info / torch.norm(mut_info, p=2)
        if strategy == 'clamp':
            return torch.clamp(mut_info, min=-self.config['mut_clamp_value'], max=self.config['mut_clamp_value'])
        return mut_info

    def loss_function(self, recon_x, x, params, K=1, **extra_loss_terms):
        nll_type = self.config['nll_type']
        nll_given = None
        if self.training and K > 1:
            nll_given = self.nll(x, recon_x, nll_type)
            for k in range(1, K):
                z_k, params_k = self.reparameterize(logits=params['logits'], labels=params.get('labels', None))
                recon_x_i = self.decode(z_k)
                nll_given = nll_given + self.nll(x, recon_x_i, nll_type)
            nll_given = nll_given / K
        kld = self.kld(params)
        mi = self.mut_info(params, x.size(0))
        kl_beta = self.compute_kl_beta([self.config['kl_beta']])[0]
        return loss_function(recon_x, x, kld, nll_type, kl_beta=kl_beta, mut_info=mi,
                             extra_loss_terms=extra_loss_terms, inv_count=self.inv_count, nll_given=nll_given)
