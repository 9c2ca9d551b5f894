"""IsotropicGaussian with the reference's interface (reparameterizers/isotropic_gaussian.py),
routed through the fused HIP kernels.

forward(): one launch does the mu/logvar split, `+ eps`, `z = e*(0.5*logvar) + mu` (NO exp,
reference :87-92), KL vs N(0,1) (:158-170) and the two means (:213-214).  The KL is kept on
the returned params dict so that kl(params) costs nothing; kl() on foreign params or with a
prior runs the standalone KL kernel.  params['gaussian']['logvar'] is produced on first read
(vae_amd/lazy.py): the training step never reads it, so its 4L bytes per sample are not written.
No CPU path: CPU tensors raise.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.distributions as D

from .. import functional as F_
from .. import rng
from ..lazy import LazyParams

EPS_TOL = 1e-6        # helpers.utils.eps(half=False): assumed, helpers is un-vendored
EPS_TOL_HALF = 1e-4   # helpers.utils.eps(half=True): assumed likewise ("parity unpinned"; tests/golden/make_golden.py
                      # stands in for helpers with the same two values)


def eps_tol(config):
    """The offset of isotropic_gaussian.py:125 (`logvar = logits[..., L:] + eps_fn(config['half'])`).  helpers is not
    vendored, so both values are this build's assumption; config['eps_tol'] overrides them."""
    if config.get('eps_tol') is not None:
        return float(config['eps_tol'])
    return EPS_TOL_HALF if config.get('half') else EPS_TOL


class IsotropicGaussian(nn.Module):
    def __init__(self, config):
        super(IsotropicGaussian, self).__init__()
        self.is_discrete = False
        self.config = config
        self.input_size = self.config['continuous_size']
        assert self.config['continuous_size'] % 2 == 0
        self.output_size = self.config['continuous_size'] // 2
        self.eps_tol = eps_tol(config)
        self.row_offset = 0       # first global row of this rank's shard (set by vae_amd.dist)
        self.mean_scale = None    # 1/(B_global*L) when sharded

    # ---- prior helpers: plain PyTorch, as in the reference (:28-75) --------------------
    def _new(self, *shape):
        dev = 'cuda' if self.config.get('cuda', True) else 'cpu'
        return torch.zeros(*shape, dtype=torch.float16 if self.config.get('half') else torch.float32, device=dev)

    def prior_params(self, batch_size, **kwargs):
        mu = self._new(batch_size, self.output_size)
        scale_var = 1.0 if 'scale_var' not in kwargs else kwargs['scale_var']
        sigma = self._new(batch_size, self.output_size) + scale_var
        return {'gaussian': {'mu': mu, 'logvar': sigma}}

    def prior_distribution(self, batch_size, **kwargs):
        params = self.prior_params(batch_size, **kwargs)
        return D.Normal(params['gaussian']['mu'], params['gaussian']['logvar'], validate_args=False)

    def prior(self, batch_size, **kwargs):
        scale_var = 1.0 if 'scale_var' not in kwargs else kwargs['scale_var']
        return self._new(batch_size, self.output_size).normal_(mean=0, std=scale_var)

    def get_reparameterizer_scalars(self):
        return {}

    # ---- the hot path ------------------------------------------------------------------
    def reparmeterize(self, logits, force=False, noise=None):
        dim_map = {2: -1, 3: -1, 4: 1}
        assert logits.dim() in dim_map, "unknown number of dims for isotropic gauss reparam"
        dim2slice = dim_map[logits.dim()]
        feature_size = logits.shape[dim2slice]
        assert feature_size % 2 == 0, "feature dimension not divisible by 2 for mu/sigma^2."
        assert feature_size // 2 == self.output_size, \
            "feature_size = {} but requested output_size = {}".format(feature_size, self.output_size)
        L = feature_size // 2
        training = bool(self.training or force)
        mu = torch.narrow(logits, dim2slice, 0, L)  # a view of logits, as in the reference

        out_shape, kl_shape = mu.shape, mu.shape[:-1]
        seed, offset = (0, 0)
        if training and noise is None:
            seed, offset = rng.next_call()
        if logits.dim() == 4:
            # [B, 2C, H, W] split on channels, KL summed over W only: the kernel walks the (b, c, h) rows of the
            # tensor where it is (vae_gauss_reparam_kl_fwd_nchw), no repacked copy
            z, kl, means = F_.gauss_reparam_kl_nchw(logits, noise=noise, eps_tol=self.eps_tol, training=training,
                                                    seed=seed, offset=offset, row0=self.row_offset,
                                                    mean_scale=None if self.mean_scale is None else
                                                    self.mean_scale / (logits.shape[2] * logits.shape[3]))
        else:
            z, _, kl, means = F_.gauss_reparam_kl(logits, noise=noise, eps_tol=self.eps_tol, training=training,
                                                  seed=seed, offset=offset, row0=self.row_offset,
                                                  mean_scale=self.mean_scale, want_logvar=False)
        z = z.reshape(out_shape) if training else mu
        g = LazyParams({'z': z, 'mu': mu})
        # the reference's own op (isotropic_gaussian.py:124-125), run only if somebody reads the entry
        eps = self.eps_tol
        g.set_lazy('logvar', lambda: torch.narrow(logits, dim2slice, L, L) + eps)
        g._vae_amd_kl = (kl.reshape(kl_shape), mu)      # consumed by kl(); invisible to dict consumers
        g._vae_amd_means = means
        return z, g

    def forward(self, logits, force=False, noise=None):
        z, gauss_params = self.reparmeterize(logits, force=force, noise=noise)
        means = gauss_params._vae_amd_means.to(logits.dtype)
        gauss_params['mu_mean'] = means[0]
        gauss_params['logvar_mean'] = means[1]
        return z, {'z': z, 'logits': logits, 'gaussian': gauss_params}

    def forward_multi(self, logits, num_samples, force=False, noise=None):
        """`num_samples` forward() calls on the same logits in one launch (the draw loops of msg.py:63-69 and
        abstract_vae.py:347-351): returns (z_list, params_list) as those loops build them.  mu / logvar / the
        cached KL are shared by all entries (they do not depend on the draw).  `noise`: [S, B, L] or a list."""
        S = int(num_samples)
        if not (self.training or force) or logits.dim() != 2:
            pairs = [self.forward(logits, force=force, noise=None if noise is None else noise[i]) for i in range(S)]
            return [p[0] for p in pairs], [p[1] for p in pairs]
        L = self.output_size
        assert logits.shape[-1] == 2 * L, "feature_size = {} but requested output_size = {}".format(logits.shape[-1], L)
        seed, offset = (0, 0) if noise is not None else rng.next_calls(S)
        z, _, kl, means = F_.gauss_reparam_multi(logits, S, noise=noise, eps_tol=self.eps_tol, seed=seed, offset=offset,
                                                 row0=self.row_offset, mean_scale=self.mean_scale, want_logvar=False)
        mu = torch.narrow(logits, -1, 0, L)
        shared = {}

        def logvar_once():      # one tensor shared by the S parameter sets, as mu is
            if 'v' not in shared:
                shared['v'] = torch.narrow(logits, -1, L, L) + self.eps_tol
            return shared['v']
        means = means.to(logits.dtype)
        z_list, params_list = [], []
        for i in range(S):
            g = LazyParams({'z': z[i], 'mu': mu})
            g.set_lazy('logvar', logvar_once)
            g._vae_amd_kl = (kl, mu)
            g['mu_mean'], g['logvar_mean'] = means[0], means[1]
            z_list.append(z[i])
            params_list.append({'z': z[i], 'logits': logits, 'gaussian': g})
        return z_list, params_list

    def kl(self, dist_a, prior=None):
        g = dist_a['gaussian']
        if prior is None:
            # the KL the forward launch already produced: kept on the params dict (single / multi / Concat forward)
            # or on the logvar tensor (fused VRNN / conditioned-prior step)
            fused = getattr(g, '_vae_amd_kl', None)
            if fused is None and not (isinstance(g, LazyParams) and g.is_pending('logvar')):
                fused = getattr(g['logvar'], '_vae_amd_kl', None)
            if fused is not None and fused[1] is g['mu']:
                return fused[0]
            return F_.gauss_kl(g['mu'], g['logvar'])
        pair = getattr(g, '_vae_amd_kl_pair', None)             # set by the fused VRNN step (vae_amd.vrnn)
        if pair is not None and pair[1] is prior['gaussian']:
            return pair[0]
        return F_.gauss_kl(g['mu'], g['logvar'], prior['gaussian']['mu'], prior['gaussian']['logvar'])

    def mutual_info(self, params, eps=None):
        kl_proxy_to_xent = F_.gauss_kl(params['gaussian']['mu'], params['gaussian']['logvar'],
                                       params['q_z_given_xhat']['gaussian']['mu'],
                                       params['q_z_given_xhat']['gaussian']['logvar'])
        return self.config['continuous_mut_info'] * kl_proxy_to_xent

    def log_likelihood(self, z, params):
        return -(0.5 * np.log(2 * np.pi) + params['gaussian']['logvar']) \
            - 0.5 * ((z - params['gaussian']['mu']) / torch.exp(params['gaussian']['logvar'])) ** 2
