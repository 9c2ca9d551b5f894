"""FlowReparameterizer with the reference's interface (reparameterizers/flow.py): constructed from the reference's
config (flow_type / continuous_size / latent_size), its "KL" reduction in the HIP kernel in O(B*F) instead of the
reference's accidental [B,B] intermediate, its MADE / coupling / BatchNormFlow tails in csrc/flowtail.hip."""
import torch
import torch.nn as nn

from .. import functional as F_


class FlowReparameterizer(nn.Module):
    """A fully flow based reparameterizer, built from the reference's own config keys (flow.py:20-49):
    config['flow_type'] in {'maf', 'maf_split', 'maf_split_glow', 'glow', 'realnvp'} with config['continuous_size']
    features, config['latent_size'] hidden units and 5 blocks, through vae_amd.flows' builders (same layer classes,
    ctor arguments and state_dict keys as flows/__init__.py, so the reference's checkpoints load; the elementwise
    tails run in csrc/flowtail.hip, the invertible 1x1 matrices of the glow variants are stock GEMMs).

    Overrides (not in the reference): config['flow_module'] -- an nn.Module whose forward(x[N,F]) returns
    (u[N,F], logdet[N,1]) and that has `.sample(num_samples, noise)` -- or config['flow_builder'],
    callable(config) -> such a module, replace the built flow."""

    def __init__(self, config):
        super(FlowReparameterizer, self).__init__()
        continuous_size = config['continuous_size']
        self.num_reads = config.get('memory_read_steps', 1)
        if config.get('flow_module') is not None:
            self.flow = config['flow_module']
        elif config.get('flow_builder') is not None:
            self.flow = config['flow_builder'](config)
        else:
            self.flow = self._build(config, continuous_size)
        self.input_size = continuous_size
        self.output_size = continuous_size
        self.inv_count = None    # 1/B_global when sharded
        self.reduce_fn = None    # all-reduce(sum) of one float when sharded (vae_amd.dist)

    @staticmethod
    def _build(config, features):
        """The five stacks of flow.py:20-49 (5 blocks each, `latent_size` hidden units), as a table: builder and the
        keyword arguments that differ from one flow_type to the next."""
        from .. import flows
        st = dict(s_activation_str='tanh', t_activation_str='relu')
        norm = dict(normalization_str=config.get('dense_normalization', 'none'), cuda=config.get('cuda', True))
        table = {
            'maf': (flows.build_maf_flow, dict(activation_str='tanh')),
            'maf_split': (flows.build_maf_split_flow, st),
            'maf_split_glow': (flows.build_maf_split_glow_flow, st),
            'realnvp': (flows.build_realnvp_flow, dict(st, **norm)),
            'glow': (flows.build_glow_flow, dict(norm, activation_str=config.get('encoder_activation', 'relu'),
                                                 layer_modifier=config.get('encoder_layer_modifier', 'none'))),
        }
        kind = config['flow_type']
        if kind not in table:
            raise ValueError("unknown flow_type '%s'" % (kind,))
        builder, extra = table[kind]
        return builder(num_inputs=features, num_hidden=config['latent_size'], num_cond_inputs=None, num_blocks=5, **extra)

    def prior(self, batch_size, scale_var=1.0, **kwargs):
        p = next(self.flow.parameters(), None)
        noise = torch.empty(batch_size, self.input_size, device=None if p is None else p.device).normal_(std=scale_var)
        return self.flow.sample(num_samples=batch_size, noise=noise)

    def get_reparameterizer_scalars(self):
        return {}

    def kl(self, dist_a, prior=None):
        return F_.flow_kl(dist_a['recon_x'], dist_a['logdet'], inv_count=self.inv_count, reduce_fn=self.reduce_fn)

    def forward(self, logits, force=False):
        """[B, F] or [B, reads, F] (flow.py:51-65): the flow runs over rows of F features; the logdet comes back as
        [B, 1] or [B, reads]."""
        if logits.dim() not in (2, 3):
            raise ValueError("FlowReparameterizer: logits of %d dimensions" % logits.dim())
        width = logits.shape[-1]
        flow, logdet = self.flow(logits.contiguous().view(-1, width))
        if logits.dim() == 3:
            reads = logits.shape[1]
            flow, logdet = flow.view(-1, reads, width), logdet.view(-1, reads)
        else:
            flow = flow.view(-1, width)
        return flow, {'recon_x': flow, 'logdet': logdet}
