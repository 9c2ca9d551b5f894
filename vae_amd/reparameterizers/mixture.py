"""Mixture (continuous + discrete side by side), reference reparameterizers/mixture.py.

The reference class is marked deprecated and cannot be constructed: its __init__ reads
self.num_continuous / self.num_discrete, which are never set (mixture.py:16-27, SURVEY Q4;
AttributeError).  This class keeps the interface and the evident intent (input_size =
num_continuous_input + num_discrete_input) so that the 'mixture' registry entry and VRNN's
`_clamp_variance` branch (vrnn.py:409-415) have something to call; its behaviour equals
ConcatReparameterizer([continuous, 'discrete']) with the reference's mixture param-dict layout."""
import torch
import torch.nn as nn

from .beta import Beta
from .gumbel import GumbelSoftmax
from .isotropic_gaussian import IsotropicGaussian


class Mixture(nn.Module):
    def __init__(self, config, is_beta=False):
        super(Mixture, self).__init__()
        self.config = config
        self.is_beta = is_beta
        self.is_discrete = False
        self.continuous = IsotropicGaussian(config) if not is_beta else Beta(config)
        self.discrete = GumbelSoftmax(config)
        self.num_discrete_input = self.discrete.input_size
        self.num_continuous_input = self.continuous.input_size
        self.num_discrete_output = self.discrete.output_size
        self.num_continuous_output = self.continuous.output_size
        self.input_size = self.num_continuous_input + self.num_discrete_input
        self.output_size = self.num_continuous_output + self.num_discrete_output

    def get_reparameterizer_scalars(self):
        return self.discrete.get_reparameterizer_scalars()

    def prior_params(self, batch_size, **kwargs):
        cont_params = self.continuous.prior_params(batch_size, **kwargs)
        disc_params = self.discrete.prior_params(batch_size, **kwargs)
        return {**disc_params, **cont_params}

    def prior_distribution(self, batch_size, **kwargs):
        return {'continuous': self.continuous.prior_distribution(batch_size, **kwargs),
                'discrete': self.discrete.prior_distribution(batch_size, **kwargs)}

    def prior(self, batch_size, **kwargs):
        disc = self.discrete.prior(batch_size, **kwargs)
        cont = self.continuous.prior(batch_size, **kwargs)
        return torch.cat([cont, disc.type(cont.dtype)], 1)

    def mutual_info(self, params):
        dinfo = self.discrete.mutual_info(params)
        cinfo = self.continuous.mutual_info(params)
        return dinfo - cinfo

    def log_likelihood(self, z, params):
        cont = self.continuous.log_likelihood(z[:, 0:self.continuous.output_size], params)
        disc = self.discrete.log_likelihood(z[:, self.continuous.output_size:], params)
        if disc.dim() < 2:
            disc = disc.unsqueeze(-1)
        return torch.cat([cont, disc], 1)

    def forward(self, logits, force=False, noise=None):
        """`noise`: optional [continuous noise, Gumbel uniforms] (each may be None).  Gaussian + Gumbel on 2-D logits
        is the pair of csrc/concat.hip: ONE launch over the [2L + K] row, the sample written contiguously; a Beta
        continuous part (its sample is ATen's) or other ranks take the two children's own launches."""
        noise = [None, None] if noise is None else list(noise)
        key = 'beta' if self.is_beta else 'gaussian'
        if not self.is_beta and logits.dim() == 2:
            from .concat_reparameterizer import fused_forward
            merged, p = fused_forward([self.continuous, self.discrete], logits, bool(self.training or force), noise)
            cont, disc = p['concat']
            return merged, {'logits': logits, 'z': merged, key: cont[key], 'discrete': disc['discrete']}
        n_c = self.num_continuous_input
        parts, params = [], {'logits': logits}
        for child, sl, n, k in ((self.continuous, slice(0, n_c), noise[0], key),
                                (self.discrete, slice(n_c, None), noise[1], 'discrete')):
            z_i, p_i = child(logits[:, sl], force=force, **({} if n is None else {'noise': n}))
            parts.append(z_i)
            params[k] = p_i[k]
        params['z'] = torch.cat(parts, -1)
        return params['z'], params

    def reparmeterize(self, logits, force=False, noise=None):
        # mixture.py:91-107 (the reference class defines no forward(); nn.Module callers go through this)
        return self.forward(logits, force=force, noise=noise)

    def kl(self, dist_a, prior=None):
        return self.continuous.kl(dist_a, prior) + self.discrete.kl(dist_a, prior)
