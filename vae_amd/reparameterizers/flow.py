"""FlowReparameterizer with the reference's interface (reparameterizers/flow.py).  The flow network
is a stock PyTorch module supplied by the host project (the reference builds it from its own
flows/ package, out of scope per SURVEY section 2); only the "KL" reduction is on the hot path and
runs in the HIP kernel, in O(B*F) instead of the reference's accidental [B,B] intermediate."""
import torch
import torch.nn as nn

from .. import functional as F_


class FlowReparameterizer(nn.Module):
    """A fully flow based reparameterizer.

    config['flow_module'] (an nn.Module whose forward(x[N,F]) returns (u[N,F], logdet[N,1]) and that
    has `.sample(num_samples, noise)`), or config['flow_builder'] (callable(config) -> such a module),
    stands in for the reference's `flows.build_*_flow` call (flow.py:26-49)."""

    def __init__(self, config):
        super(FlowReparameterizer, self).__init__()
        continuous_size = config['continuous_size']
        self.num_reads = config.get('memory_read_steps', 1)
        if config.get('flow_module') is not None:
            self.flow = config['flow_module']
        elif config.get('flow_builder') is not None:
            self.flow = config['flow_builder'](config)
        else:
            raise ValueError("FlowReparameterizer needs config['flow_module'] or config['flow_builder']: the flow "
                             "networks are stock PyTorch modules of the host project (reference flows/)")
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
