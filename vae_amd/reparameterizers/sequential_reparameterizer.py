"""SequentialReparameterizer with the reference's interface
(reparameterizers/sequential_reparameterizer.py): reparam0 -> dense -> reparam1 -> ...; KLs summed
over params['sequential'].  The dense projections are stock PyTorch (the reference takes them from
helpers.layers.get_encoder, un-vendored): config['dense_builder'](input_size, output_size, name) may
supply the host project's builder, default one nn.Linear."""
import torch.nn as nn


class SequentialReparameterizer(nn.Module):
    def __init__(self, reparam_strs, config):
        super(SequentialReparameterizer, self).__init__()
        self.config = config
        self.reparam_strs = reparam_strs
        self.reparameterizers = nn.ModuleList(self._generate_reparameterizers())
        self.input_size = self.reparameterizers[0].input_size
        self.output_size = self.reparameterizers[-1][-1].output_size \
            if len(self.reparameterizers) > 1 else self.reparameterizers[-1].output_size

    def _get_dense_net(self, input_size, output_size, name='dense'):
        builder = self.config.get('dense_builder')
        if builder is not None:
            return builder(input_size, output_size, name)
        return nn.Linear(input_size, output_size)

    def _append_inter_projection(self, reparam, input_size, output_size, name):
        return nn.Sequential(self._get_dense_net(input_size, output_size, name), reparam)

    def _generate_reparameterizers(self):
        from . import get_reparameterizer
        reparam_list = [get_reparameterizer(reparam)(config=self.config) for reparam in self.reparam_strs]
        input_size = reparam_list[0].output_size
        for i in range(1, len(reparam_list)):
            reparam_list[i] = self._append_inter_projection(reparam_list[i], input_size=input_size,
                                                            output_size=reparam_list[i].input_size,
                                                            name='reparam_proj{}'.format(i))
            input_size = reparam_list[i][-1].output_size
        return reparam_list

    @staticmethod
    def _leaf(reparam):
        return reparam[-1] if isinstance(reparam, nn.Sequential) else reparam

    def get_reparameterizer_scalars(self):
        reparam_scalar_map = {}
        for i, reparam in enumerate(self.reparameterizers):
            for k, v in self._leaf(reparam).get_reparameterizer_scalars().items():
                reparam_scalar_map["{}{}".format(i, k)] = v
        return reparam_scalar_map

    def mutual_info(self, dists):
        dist_params, mi = dists['concat'], None   # sic: the reference reads 'concat' here (:97)
        for param, reparam in zip(dist_params, self.reparameterizers):
            m = self._leaf(reparam).mutual_info(param)
            mi = m if mi is None else mi + m
        return mi

    def kl(self, dists, priors=None):
        dist_params = dists['sequential']
        priors = [None for _ in range(len(dist_params))] if priors is None else priors
        assert len(priors) == len(dist_params)
        kl = None
        for param, prior, reparam in zip(dist_params, priors, self.reparameterizers):
            k = self._leaf(reparam).kl(param, prior)
            kl = k if kl is None else kl + k
        return kl

    def forward(self, logits, force=False, noise=None):
        """`noise`: optional list with one injected noise tensor (or None) per child."""
        from . import is_module_a_reparameterizer
        assert force is False, "force not implemented for sequential reparameterizer"
        original_logits = logits
        noise = [None] * len(self.reparameterizers) if noise is None else noise
        kw = {} if noise[0] is None else {'noise': noise[0]}
        logits, params = self.reparameterizers[0](logits, force=force, **kw)
        params_list = [{**params, 'logits': logits}]
        for i, reparam in enumerate(self.reparameterizers[1:], start=1):
            for layer in reparam:
                if is_module_a_reparameterizer(layer):
                    kw = {} if noise[i] is None else {'noise': noise[i]}
                    logits, params = layer(logits, force=force, **kw)
                else:
                    logits = layer(logits)
            params_list.append({**params, 'logits': logits})
        return logits, {'logits': original_logits, 'sequential': params_list}

    def prior(self, batch_size, **kwargs):
        prior = self.reparameterizers[0].prior(batch_size)
        if len(self.reparameterizers) > 1:
            for reparam in self.reparameterizers[1:]:
                prior, _ = reparam(prior)
        return prior
