#!/usr/bin/env python3
"""Host-side cost of the module path for the C2-shaped step at a small batch (cProfile, sorted by own time)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_amd.reparameterizers import IsotropicGaussian  # noqa: E402
from vae_amd.elbo import loss_function  # noqa: E402

dev = torch.device('cuda:0')
B, L = 1024, 128
g = IsotropicGaussian({'continuous_size': 2 * L, 'cuda': True, 'half': False}).train()
lg = torch.randn(B, 2 * L, device=dev, requires_grad=True)
x = (torch.rand(B, 1, 28, 28, device=dev) < 0.13).float()
rc = torch.randn(B, 1, 28, 28, device=dev, requires_grad=True)
gzz = torch.randn(B, L, device=dev) / B


def step():
    z, params = g(lg)
    out = loss_function(rc, x, g.kl(params), 'bernoulli')
    torch.autograd.backward([out['loss_mean'], z], [None, gzz])
    lg.grad = None
    rc.grad = None


for _ in range(30):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(40)
