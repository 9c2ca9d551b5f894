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
            from .. import flows
            flow_type, latent_size = config['flow_type'], config['latent_size']
            if flow_type == 'maf':
                self.flow = flows.build_maf_flow(num_inputs=continuous_size, num_hidden=latent_size,
                                                 num_cond_inputs=None, num_blocks=5, activation_str='tanh')
            elif flow_type == 'maf_split':
                self.flow = flows.build_maf_split_flow(num_inputs=continuous_size, num_hidden=latent_size,
                                                       num_cond_inputs=None, num_blocks=5, s_activation_str='tanh',
                                                       t_activation_str='relu')
            elif flow_type == 'realnvp':
                self.flow = flows.build_realnvp_flow(num_inputs=continuous_size, num_hidden=latent_size,
                                                     num_cond_inputs=None, num_blocks=5, s_activation_str='tanh',
                                                     t_activation_str='relu',
                                                     normalization_str=config.get('dense_normalization', 'none'),
                                                     cuda=config.get('cuda', True))
            elif flow_type == 'maf_split_glow':
                self.flow = flows.build_maf_split_glow_flow(num_inputs=continuous_size, num_hidden=latent_size,
                                                            num_cond_inputs=None, num_blocks=5,
                                                            s_activation_str='tanh', t_activation_str='relu')
            elif flow_type == 'glow':
                self.flow = flows.build_glow_flow(num_inputs=continuous_size, num_hidden=latent_size,
                                                  num_cond_inputs=None, num_blocks=5,
                                                  activation_str=config.get('encoder_activation', 'relu'),
                                                  normalization_str=config.get('dense_normalization', 'none'),
                                                  layer_modifier=config.get('encoder_layer_modifier', 'none'),
                                                  cuda=config.get('cuda', True))
            else:
                raise ValueError("unknown flow_type '%s'" % (flow_type,))
        self.input_size = continuous_size
        self.output_size = continuous_size
        self.inv_count = None    # 1/B_global when sharded
        self.reduce_fn = None    # all-reduce(sum) of one float when sharded (vae_amd.dist)

    def prior(self, batch_size, scale_var=1.0, **kwargs):
        p = next(self.flow.parameters(), None)
        noise = torch.empty(batch_size, self.input_size, device=None if p is None else p.device).normal_(std=scale_var)
        return self.flow.sample(num_samples=batch_size, noise=noise)

    def get_reparameterizer_scalars(self):
        return {}

    def kl(self, dist_a, prior=None):
        return F_.flow_kl(dist_a['recon_x'], dist_a['logdet'], inv_count=self.inv_count, reduce_fn=self.reduce_fn)

    def forward(self, logits, force=False):
        if logits.dim() == 3:
            batch_size, num_reads, input_size = logits.shape
            flow, logdet = self.flow(logits.contiguous().view(-1, input_size))
            flow = flow.view([-1, num_reads, input_size])
            logdet = logdet.view([-1, num_reads])
        elif logits.dim() == 2:
            batch_size, input_size = logits.shape
            flow, logdet = self.flow(logits.contiguous().view(-1, input_size))
            flow = flow.view([-1, input_size])
        return flow, {'recon_x': flow, 'logdet': logdet}
