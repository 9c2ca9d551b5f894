"""ConcatReparameterizer with the reference's interface (reparameterizers/concat_reparameterizer.py):
children get narrow() views of the logits (no copies: the kernels take a row stride), their outputs
are concatenated, their KLs summed."""
import numpy as np
import torch
import torch.nn as nn


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
        params_list = []
        reparameterized = []
        dim_map = {2: -1, 3: -1, 4: 1}
        assert logits.dim() in dim_map, "unknown number of dims for concat reparam"
        dim2slice = dim_map[logits.dim()]
        noise = [None] * len(self.reparameterizers) if noise is None else noise
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
