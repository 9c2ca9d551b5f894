#!/usr/bin/env python3
"""Host-side cost of the module path (cProfile over the C3 Concat step; GPU work is asynchronous)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_amd.reparameterizers import ConcatReparameterizer  # noqa: E402

B, L, K = int(os.environ.get('PB', 65536)), 128, 256
dev = torch.device('cuda:0')
logits = torch.randn(B, 2 * L + K, device=dev, requires_grad=True)
gz = torch.randn(B, L + K, device=dev) / B
gkl = torch.full((B,), 1.0 / B, device=dev)
m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'],
                          {'continuous_size': 2 * L, 'discrete_size': K, 'cuda': True, 'half': False}).train()


def step():
    z, params = m(logits)
    kl = m.kl(params)
    torch.autograd.backward([z, kl], [gz, gkl])
    logits.grad = None


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
