"""ConcatReparameterizer with the reference's interface (reparameterizers/concat_reparameterizer.py):
children get narrow() views of the logits (no copies: the kernels take a row stride), their outputs
are concatenated, their KLs summed."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as F_
from .. import rng
from ..lazy import LazyParams, hard_from_soft_max, hard_from_soft_threshold
from .bernoulli import Bernoulli
from .gumbel import GumbelSoftmax
from .isotropic_gaussian import IsotropicGaussian

_FUSABLE = (IsotropicGaussian, GumbelSoftmax, Bernoulli)


def fused_forward(children, logits, training, noise):
    """All children are kernel-backed leaves on 2-D logits: one autograd node; every child reads its
    strided slice of `logits`, writes its sample into the concatenated output and (backward) its gradient
    into one shared d(logits) buffer -- a (Gaussian, Gumbel) pair in ONE launch (csrc/concat.hip).  Returns the
    concatenated sample and {'logits', 'concat': [per-child params]} as ConcatReparameterizer.forward does; Mixture
    re-labels the same entries."""
    specs, seeds = [], []
    for child, n in zip(children, noise):
        if isinstance(child, IsotropicGaussian):
            specs.append(('gauss', child.output_size, child.eps_tol, child.mean_scale))
        else:
            child.anneal()   # host-side tau schedule, before the draw (gumbel.py:326, bernoulli.py:198)
            specs.append(('gumbel' if isinstance(child, GumbelSoftmax) else 'bern', child.output_size,
                          float(child.tau)))
        seeds.append((0, 0) if (n is not None or (specs[-1][0] == 'gauss' and not training)) else rng.next_call())
    row0 = children[0].row_offset
    outs = F_.concat_leaves(logits, specs, training, noise, seeds, row0=row0)
    zcat, k = outs[0], 1
    params_list, i_off, o_off = [], 0, 0
    for child, spec in zip(children, specs):
        kind, n_out = spec[0], spec[1]
        z_i = zcat[:, o_off:o_off + n_out]
        if kind == 'gauss':
            logits_i = logits[:, i_off:i_off + 2 * n_out]
            mu = logits_i[:, :n_out]
            kl, means = outs[k + 1], outs[k + 2]
            g = LazyParams({'z': z_i, 'mu': mu})
            g.set_lazy('logvar', lambda li=logits_i, n=n_out, e=spec[2]: li[:, n:] + e)   # on first read only
            g._vae_amd_kl = (kl, mu)
            if means.numel():
                means = means.to(logits.dtype)
                g['mu_mean'], g['logvar_mean'] = means[0], means[1]
            else:
                # the one-launch row kernel takes no grid-wide sums: the two diagnostics are the reference's own
                # ops (isotropic_gaussian.py:213-214), run if and when somebody reads them
                ms = child.mean_scale
                scale = (lambda t: torch.mean(t)) if ms is None else (lambda t, c=ms: torch.sum(t) * c)
                g.set_lazy('mu_mean', lambda m=mu, f=scale: f(m))
                g.set_lazy('logvar_mean', lambda gg=g, f=scale: f(gg['logvar']))
            params = {'z': z_i, 'logits': logits_i, 'gaussian': g}
            k, i_off = k + 3, i_off + 2 * n_out
        else:
            logits_i = logits[:, i_off:i_off + n_out]
            other = outs[k]
            y, z_hard = (z_i, None) if training else (other, z_i)
            d = LazyParams()
            if training:   # the launch wrote y only (see GumbelSoftmax.forward / Bernoulli.forward)
                d.set_lazy('z_hard', (lambda yy=y: hard_from_soft_max(yy)) if kind == 'gumbel' else
                           (lambda yy=y: hard_from_soft_threshold(yy)))
            else:
                d['z_hard'] = z_hard
            d['logits'] = logits_i
            if kind == 'gumbel':
                if training:
                    d.set_lazy('log_q_z', lambda li=logits_i: F_.log_softmax_rows(li))
                else:
                    d['log_q_z'] = outs[k + 1]
                d['tau_scalar'] = child.tau
                params = {'z': y, 'discrete': d}
                k += 2
            else:
                d['tau_scalar'] = child.tau
                params = {'z': y, 'logits': logits_i, 'discrete': d}
                k += 1
            child.iteration += 1
            i_off += n_out
        params_list.append({**params, 'logits': logits_i})
        o_off += n_out
    return zcat, {'logits': logits, 'concat': params_list}


class ConcatReparameterizer(nn.Module):
    def __init__(self, reparam_strs, config):
        super(ConcatReparameterizer, self).__init__()
        from . import get_reparameterizer
        self.config = config
        self.reparam_strs = reparam_strs
        self.reparameterizers = nn.ModuleList([get_reparameterizer(r)(config=self.config) for r in reparam_strs])
        self.input_size = int(np.sum([r.input_size for r in self.reparameterizers]))
        self.output_size = int(np.sum([r.output_size for r in self.reparameterizers]))
        self._input_sizing = [0] + list(np.cumsum([r.input_size for r in self.reparameterizers]))

    def get_reparameterizer_scalars(self):
        reparam_scalar_map = {}
        for i, reparam in enumerate(self.reparameterizers):
            for k, v in reparam.get_reparameterizer_scalars().items():
                reparam_scalar_map["{}{}".format(i, k)] = v
        return reparam_scalar_map

    def mutual_info(self, dists):
        dist_params, mi = dists['concat'], None
        for param, reparam in zip(dist_params, self.reparameterizers):
            mi = reparam.mutual_info(param) if mi is None else mi + reparam.mutual_info(param)
        return mi

    def kl(self, dists, priors=None):
        dist_params = dists['concat']
        priors = [None for _ in range(len(dist_params))] if priors is None else priors
        assert len(priors) == len(dist_params)
        kl = None
        for param, prior, reparam in zip(dist_params, priors, self.reparameterizers):
            kl = reparam.kl(param, prior) if kl is None else kl + reparam.kl(param, prior)
        return kl

    def forward(self, logits, force=False, noise=None):
        """`noise`: optional list with one injected noise tensor (or None) per child."""
        noise = [None] * len(self.reparameterizers) if noise is None else noise
        if logits.dim() == 2 and all(isinstance(r, _FUSABLE) for r in self.reparameterizers):
            return fused_forward(list(self.reparameterizers), logits, bool(self.training or force), noise)
        params_list = []
        reparameterized = []
        dim_map = {2: -1, 3: -1, 4: 1}
        assert logits.dim() in dim_map, "unknown number of dims for concat reparam"
        dim2slice = dim_map[logits.dim()]
        for i, (begin, end) in enumerate(zip(self._input_sizing, self._input_sizing[1:])):
            begin, end = int(begin), int(end)
            logits_i = torch.narrow(logits, dim2slice, begin, end - begin)
            kw = {} if noise[i] is None else {'noise': noise[i]}
            reparameterized_i, params = self.reparameterizers[i](logits_i, force=force, **kw)
            reparameterized.append(reparameterized_i)
            params_list.append({**params, 'logits': logits[:, begin:end]})
        return torch.cat(reparameterized, -1), {'logits': logits, 'concat': params_list}

    def prior(self, batch_size, **kwargs):
        return torch.cat([r.prior(batch_size) for r in self.reparameterizers], -1)
