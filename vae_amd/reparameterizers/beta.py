"""Beta with the reference's interface (reparameterizers/beta.py).  The concentration split, the
KL vs Beta(1.1, 1.1) and the two means are one fused launch; the sample itself stays
torch.distributions.Beta (ATen's rejection sampler with its implicit-reparameterization gradient:
not injectable), whose gradient flows back into the fused kernel's backward."""
import torch
import torch.nn as nn
import torch.distributions as D

from .. import functional as F_


class Beta(nn.Module):
    def __init__(self, config):
        super(Beta, self).__init__()
        self.is_discrete = False
        self.config = config
        self.input_size = self.config['continuous_size']
        assert self.config['continuous_size'] % 2 == 0
        self.output_size = self.config['continuous_size'] // 2
        self.mean_scale = None

    def get_reparameterizer_scalars(self):
        return {}

    def _new(self, *shape):
        dev = 'cuda' if self.config.get('cuda', True) else 'cpu'
        return torch.zeros(*shape, dtype=torch.float16 if self.config.get('half') else torch.float32, device=dev)

    def prior(self, batch_size, **kwargs):
        conc1 = self._new(batch_size, self.output_size) + 1 / 3
        conc2 = self._new(batch_size, self.output_size) + 1 / 3
        return D.Beta(conc1, conc2, validate_args=False).sample()

    def _reparametrize_beta(self, conc1, conc2, force=False):
        if self.training or force:
            return D.Beta(conc1, conc2, validate_args=False).rsample(), {'conc1': conc1, 'conc2': conc2}
        # eval still samples (beta.py:75-76)
        return D.Beta(conc1, conc2, validate_args=False).sample(), {'conc1': conc1, 'conc2': conc2}

    def reparmeterize(self, logits, force=False):
        dim_map = {2: -1, 3: -1, 4: 1}
        assert logits.dim() in dim_map, "unknown number of dims for isotropic gauss reparam"
        dim2slice = dim_map[logits.dim()]
        feature_size = logits.shape[dim2slice]
        assert feature_size % 2 == 0, "feature dimension not divisible by 2 for mu/sigma^2."
        assert feature_size // 2 == self.output_size, \
            "feature_size = {} but requested output_size = {}".format(feature_size, self.output_size)
        L = feature_size // 2
        if logits.dim() == 4:
            W = logits.shape[-1]
            a = torch.narrow(logits, 1, 0, L)
            rows = torch.cat([a.reshape(-1, W), torch.narrow(logits, 1, L, L).reshape(-1, W)], -1)
            out_shape = a.shape
        else:
            rows = logits
            out_shape = logits.shape[:-1] + (L,)
        conc1, conc2, kl, means = F_.beta_split_kl(rows, mean_scale=self.mean_scale)
        conc1, conc2 = conc1.reshape(out_shape), conc2.reshape(out_shape)
        conc2._vae_amd_kl = (kl.reshape(out_shape[:-1]), conc1)
        conc2._vae_amd_means = means
        return self._reparametrize_beta(conc1, conc2, force=force)

    def kl(self, dist_a, prior=None):
        b = dist_a['beta']
        if prior is None:
            fused = getattr(b['conc2'], '_vae_amd_kl', None)
            if fused is not None and fused[1] is b['conc1']:
                return fused[0]
            return F_.beta_kl(b['conc1'], b['conc2'])
        return F_.beta_kl(b['conc1'], b['conc2'], prior['beta']['conc1'], prior['beta']['conc2'])

    def mutual_info(self, params, eps=1e-9):
        kl_proxy_to_xent = F_.beta_kl(params['q_z_given_xhat']['beta']['conc1'],
                                      params['q_z_given_xhat']['beta']['conc2'],
                                      params['beta']['conc1'], params['beta']['conc2'])
        return self.config['continuous_mut_info'] * kl_proxy_to_xent

    def log_likelihood(self, z, params):
        return D.Beta(params['beta']['conc1'], params['beta']['conc2'], validate_args=False).log_prob(z)

    def forward(self, logits, force=False):
        z, beta_params = self.reparmeterize(logits, force=force)
        means = beta_params['conc2']._vae_amd_means.to(logits.dtype)
        beta_params['conc1_mean'] = means[0]
        beta_params['conc2_mean'] = means[1]
        return z, {'z': z, 'logits': logits, 'beta': beta_params}
