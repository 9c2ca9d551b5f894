#!/usr/bin/env python3
"""C3 module step for rocprofv3: ConcatReparameterizer(IsotropicGaussian L=128, GumbelSoftmax K=256) at B=65536,
forward + kl + backward with an upstream gradient on z given directly (no torch ops in the step besides autograd's
own).  rocprofv3 --kernel-trace --stats -- python3 scripts/profile_concat.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_amd.reparameterizers import ConcatReparameterizer  # noqa: E402

B, L, K = 65536, 128, 256
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
logits = torch.randn(B, 2 * L + K, generator=g, device=dev, requires_grad=True)
gz = torch.randn(B, L + K, generator=g, device=dev) / B
gkl = torch.full((B,), 1.0 / B, device=dev)
m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'],
                          {'continuous_size': 2 * L, 'discrete_size': K, 'cuda': True, 'half': False}).train()


def step():
    z, params = m(logits)          # in-kernel Philox noise (the classes' default)
    kl = m.kl(params)
    torch.autograd.backward([z, kl], [gz, gkl])
    logits.grad = None


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 50
for _ in range(N):
    step()
torch.cuda.synchronize()
print('concat module step: %.4f ms' % ((time.perf_counter() - t0) / N * 1e3))
