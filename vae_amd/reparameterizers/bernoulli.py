"""Bernoulli (relaxed) with the reference's interface (reparameterizers/bernoulli.py), routed through
the fused HIP kernels (sample + straight-through hard in one launch, KL in one launch)."""
import math
import numpy as np
import torch
import torch.nn as nn
import torch.distributions as D

from .. import functional as F_
from .. import rng
from ..lazy import LazyParams, hard_from_soft_threshold


class Bernoulli(nn.Module):
    def __init__(self, config, dim=-1):
        super(Bernoulli, self).__init__()
        self.is_discrete = True
        self.dim = dim
        self.iteration = 0
        self.config = config
        self.input_size = self.config['discrete_size']
        self.output_size = self.config['discrete_size']
        self._setup_anneal_params()
        self.row_offset = 0

    def _new(self, *shape):
        dev = 'cuda' if self.config.get('cuda', True) else 'cpu'
        return torch.zeros(*shape, dtype=torch.float16 if self.config.get('half') else torch.float32, device=dev)

    def prior_params(self, batch_size, **kwargs):
        uniform_probs = self._new(batch_size, self.output_size)
        uniform_probs += 0.5
        return {'discrete': {'logits': D.Bernoulli(probs=uniform_probs).logits}}

    def prior_distribution(self, batch_size, **kwargs):
        params = self.prior_params(batch_size, **kwargs)
        return torch.distributions.Bernoulli(logits=params['discrete']['logits'])

    def prior(self, batch_size, **kwargs):
        return self.prior_distribution(batch_size, **kwargs).sample()

    def _setup_anneal_params(self):
        self.anneal_rate = 3e-6
        self.tau0 = 1.0
        self.min_temp = 1e-6
        self.max_temp = 1.0
        self.tau = 1.0
        self.last_epoch = self.config['epochs']
        self.clip_min = 1e-8

    def get_reparameterizer_scalars(self):
        return {'tau_scalar': self.tau}

    def anneal(self, anneal_interval=10):
        if self.training and self.iteration > 0 and self.iteration % anneal_interval == 0:
            rate = -self.anneal_rate * self.iteration
            self.tau = np.maximum(self.tau0 * np.exp(rate), self.min_temp)

    def cosine_anneal(self):
        if self.training and self.iteration > 0:
            updated_tau = self.min_temp + (self.tau - self.min_temp) \
                * (1 + math.cos(math.pi * -1 / self.last_epoch)) / 2
            self.tau = np.clip(updated_tau, self.clip_min, self.max_temp)

    def reparmeterize(self, logits, noise=None, want_hard=True):
        seed, offset = (0, 0) if noise is not None else rng.next_call()
        return F_.relaxed_bernoulli(logits, unif=noise, tau=float(self.tau), seed=seed, offset=offset,
                                    row0=self.row_offset, want_hard=want_hard)

    def mutual_info_analytic(self, params, eps=1e-9):
        raise NotImplementedError

    def mutual_info_monte_carlo(self, params, eps=1e-9):
        raise NotImplementedError

    def mutual_info(self, params, eps=1e-9):
        raise NotImplementedError

    def kl(self, dist_a, prior=None, eps=1e-6):
        assert eps == 1e-6, "the kernel hard-codes the reference's eps"
        if prior is None:
            return F_.bernoulli_kl(dist_a['discrete']['logits'])
        return F_.bernoulli_kl(dist_a['discrete']['logits'], prior['discrete']['logits'])

    def log_likelihood(self, z, params):
        return D.Bernoulli(logits=params['discrete']['logits']).log_prob(z)

    def forward(self, logits, force=False, noise=None):
        self.anneal()
        self.iteration += 1
        if not (self.training or force):
            z, z_hard = self.reparmeterize(logits, noise=noise)
            params = {'z_hard': z_hard, 'logits': logits, 'tau_scalar': self.tau}
            return z_hard, {'z': z, 'logits': logits, 'discrete': params}
        # training: only the relaxed sample is written; z_hard is produced on first read (vae_amd/lazy.py)
        z, _ = self.reparmeterize(logits, noise=noise, want_hard=False)
        params = LazyParams()
        params.set_lazy('z_hard', lambda: hard_from_soft_threshold(z))
        params['logits'] = logits
        params['tau_scalar'] = self.tau
        return z, {'z': z, 'logits': logits, 'discrete': params}
