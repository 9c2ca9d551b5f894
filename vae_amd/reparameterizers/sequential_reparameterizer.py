"""SequentialReparameterizer with the reference's interface
(reparameterizers/sequential_reparameterizer.py:10-164): a chain  leaf0 -> dense -> leaf1 -> dense -> leaf2 ...
whose sample is the last leaf's output and whose KL is the sum of the leaves' KLs over params['sequential'].

Layout contract kept for drop-in use and checkpoints: `self.reparameterizers` is an nn.ModuleList whose first entry
is the first leaf and whose later entries are nn.Sequential(dense, leaf) (state_dict keys `reparameterizers.i.0.*`
for the projection, `reparameterizers.i.1.*` for the leaf: tests/golden/sequential.pt is the reference's own
state_dict and loads as is); attributes `reparam_strs`, `input_size`, `output_size`, `config`.

The dense projections are stock PyTorch (the reference takes them from helpers.layers.get_encoder, un-vendored):
config['dense_builder'](input_size, output_size, name) may supply the host project's builder, default one nn.Linear.
Every leaf's sample and KL runs in its own fused kernels (Gaussian: csrc/gauss.hip, Flow: csrc/flowtail.hip +
csrc/flow.hip); `forward(..., noise=[...])` injects one noise tensor (or None) per leaf.
"""
import torch.nn as nn


class SequentialReparameterizer(nn.Module):
    def __init__(self, reparam_strs, config):
        super(SequentialReparameterizer, self).__init__()
        from . import get_reparameterizer
        self.config = config
        self.reparam_strs = reparam_strs
        leaves = [get_reparameterizer(name)(config=config) for name in reparam_strs]
        stages = [leaves[0]]
        for i, leaf in enumerate(leaves[1:], start=1):
            # leaf i reads what leaf i-1 emits, through a learned projection (reference :38-69)
            proj = self._get_dense_net(leaves[i - 1].output_size, leaf.input_size, name='reparam_proj{}'.format(i))
            stages.append(nn.Sequential(proj, leaf))
        self.reparameterizers = nn.ModuleList(stages)
        self.input_size = leaves[0].input_size
        self.output_size = leaves[-1].output_size

    def _get_dense_net(self, input_size, output_size, name='dense'):
        builder = self.config.get('dense_builder')
        return builder(input_size, output_size, name) if builder is not None else nn.Linear(input_size, output_size)

    def _chain(self):
        """(projection modules in order, leaf) per stage; stage 0 has no projection."""
        from . import is_module_a_reparameterizer
        for stage in self.reparameterizers:
            if isinstance(stage, nn.Sequential):
                mods = list(stage)
                split = [is_module_a_reparameterizer(m) for m in mods]
                assert split[-1] and not any(split[:-1]), "a stage is projection layers followed by one reparameterizer"
                yield mods[:-1], mods[-1]
            else:
                yield [], stage

    def get_reparameterizer_scalars(self):
        return {"{}{}".format(i, k): v for i, (_, leaf) in enumerate(self._chain())
                for k, v in leaf.get_reparameterizer_scalars().items()}

    def mutual_info(self, dists):
        # the reference reads the parameter list under 'concat' here (:97), not 'sequential': kept
        terms = [leaf.mutual_info(p) for p, (_, leaf) in zip(dists['concat'], self._chain())]
        return sum(terms[1:], terms[0]) if terms else None

    def kl(self, dists, priors=None):
        """Sum over the leaves of leaf.kl(params_i, prior_i) (reference :104-122)."""
        params = dists['sequential']
        priors = [None] * len(params) if priors is None else priors
        assert len(priors) == len(params)
        terms = [leaf.kl(p, q) for p, q, (_, leaf) in zip(params, priors, self._chain())]
        return sum(terms[1:], terms[0])

    def forward(self, logits, force=False, noise=None):
        """-> (last leaf's sample, {'logits': the input, 'sequential': [leaf params + that leaf's output under
        'logits', ...]}) (reference :124-148).  `noise`: optional list, one tensor or None per leaf."""
        assert force is False, "force not implemented for sequential reparameterizer"
        stages = list(self._chain())
        noise = [None] * len(stages) if noise is None else noise
        h, collected = logits, []
        for (projections, leaf), n in zip(stages, noise):
            for p in projections:
                h = p(h)
            h, params = leaf(h, force=force) if n is None else leaf(h, force=force, noise=n)
            collected.append({**params, 'logits': h})
        return h, {'logits': logits, 'sequential': collected}

    def prior(self, batch_size, **kwargs):
        """A draw from the first leaf's prior pushed through the rest of the chain (reference :150-163)."""
        stages = list(self._chain())
        h = stages[0][1].prior(batch_size)
        for projections, leaf in stages[1:]:
            for p in projections:
                h = p(h)
            h, _ = leaf(h)
        return h
