"""GumbelSoftmax with the reference's interface (reparameterizers/gumbel.py), routed through the
fused HIP kernel: log_softmax, Gumbel noise, temperature softmax, the == rowmax hard mask
(multi-hot on ties) and the straight-through sample in one launch.  The reference draws U with
torch.rand on the CPU and copies [B,K] to the device every step (:269-273); here U is either
injected (`noise=`) or drawn by Philox inside the kernel.  In training only the soft sample is written; `z_hard` and
`log_q_z` are produced on first read (vae_amd/lazy.py), with the values the eager launch gives."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.distributions as D

from .. import functional as F_
from .. import rng
from ..lazy import LazyParams, hard_from_soft_max


class GumbelSoftmax(nn.Module):
    def __init__(self, config, dim=-1):
        super(GumbelSoftmax, self).__init__()
        self.is_discrete = True
        self._setup_anneal_params()
        self.dim = dim
        self.iteration = 0
        self.config = config
        self.input_size = self.config['discrete_size']
        self.output_size = self.config['discrete_size']
        self.row_offset = 0

    def _new(self, *shape):
        dev = 'cuda' if self.config.get('cuda', True) else 'cpu'
        return torch.zeros(*shape, dtype=torch.float16 if self.config.get('half') else torch.float32, device=dev)

    def prior_params(self, batch_size, **kwargs):
        uniform_probs = self._new(batch_size, self.output_size)
        uniform_probs += 1.0 / self.output_size
        return {'discrete': {'logits': D.OneHotCategorical(probs=uniform_probs).logits}}

    def prior_distribution(self, batch_size, **kwargs):
        params = self.prior_params(batch_size, **kwargs)
        return D.OneHotCategorical(logits=params['discrete']['logits'])

    def prior(self, batch_size, **kwargs):
        return self.prior_distribution(batch_size, **kwargs).sample()

    def get_reparameterizer_scalars(self):
        return {'tau_scalar': self.tau}

    def _setup_anneal_params(self):
        self.tau, self.tau0 = 1.0, 1.0
        self.anneal_rate = 3e-6
        self.min_temp = 0.5

    def anneal(self, anneal_interval=10):
        # host-side scalar state, identical update rule (gumbel.py:91-108); not checkpointed there either
        if self.training and self.iteration > 0 and self.iteration % anneal_interval == 0:
            rate = -self.anneal_rate * self.iteration
            self.tau = np.maximum(self.tau0 * np.exp(rate), self.min_temp)

    def reparmeterize(self, logits, noise=None, want=(True, True, True)):
        assert self.dim in (-1, logits.dim() - 1), "the HIP path reduces over the last dim"
        seed, offset = (0, 0) if noise is not None else rng.next_call()
        return F_.gumbel_softmax(logits, unif=noise, tau=float(self.tau), seed=seed, offset=offset,
                                 row0=self.row_offset, want=want)

    def mutual_info_monte_carlo(self, params, eps=1e-9):
        # gumbel.py:153-157: sum p_z log q(z|xhat) - sum p_z log q(z|x) with p_z = softmax(logits) = exp(log_q_z),
        # i.e. -KL(q(z|x) || q(z|xhat)): one launch of the categorical-KL kernel instead of six torch ops.
        return -F_.categorical_kl(params['discrete']['log_q_z'], params['q_z_given_xhat']['discrete']['log_q_z'])

    def mutual_info_analytic(self, params, eps=1e-9):
        # gumbel.py:126-141 (not the live form: mutual_info() returns the Monte-Carlo one): plain PyTorch
        targets = torch.argmax(F.softmax(params['discrete']['logits'], -1), dim=-1).long()
        crossent_loss = F.cross_entropy(input=params['q_z_given_xhat']['discrete']['logits'], target=targets,
                                        reduction='none')
        ent_loss = -torch.sum(D.OneHotCategorical(logits=params['discrete']['z_hard'],
                                                  validate_args=False).entropy(), -1)
        return ent_loss + crossent_loss

    def mutual_info(self, params, eps=1e-9):
        return self.config['discrete_mut_info'] * self.mutual_info_monte_carlo(params)

    def kl(self, dist_a, prior=None):
        if prior is None:
            # the reference builds the "prior" from the same log_q_z (gumbel.py:243): KL(q||q), exactly 0
            # in value and gradient; no launch is spent on it (nor on producing log_q_z).
            ref = dist_a['discrete']['logits']
            return torch.zeros(ref.shape[:-1], dtype=ref.dtype, device=ref.device)
        return F_.categorical_kl(dist_a['discrete']['log_q_z'], prior['discrete']['log_q_z'])

    def log_likelihood(self, z, params):
        return D.OneHotCategorical(logits=params['discrete']['logits']).log_prob(z)

    def forward(self, logits, force=False, noise=None):
        self.anneal()  # anneal first
        soft = bool(self.training or force)
        self.iteration += 1
        if not soft:
            z, z_hard, log_q_z = self.reparmeterize(logits, noise=noise)
            params = {'z_hard': z_hard, 'logits': logits, 'log_q_z': log_q_z, 'tau_scalar': self.tau}
            return z_hard, {'z': z, 'discrete': params}
        # training: the step reads the soft sample only, so the launch writes nothing else; z_hard and log_q_z keep
        # their places in the params dict and are produced on first read (12K bytes per sample instead of 20K)
        z, _, _ = self.reparmeterize(logits, noise=noise, want=(True, False, False))
        params = LazyParams()
        params.set_lazy('z_hard', lambda: hard_from_soft_max(z))
        params['logits'] = logits
        params.set_lazy('log_q_z', lambda: F_.log_softmax_rows(logits))
        params['tau_scalar'] = self.tau
        return z, {'z': z, 'discrete': params}
