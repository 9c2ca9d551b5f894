#!/usr/bin/env python3
"""Wall time per module-level step at a batch small enough that the GPU is never the limit: the host cost of the
drop-in classes (Python + autograd + allocator + ctypes) per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_amd.reparameterizers import ConcatReparameterizer, IsotropicGaussian  # noqa: E402
from vae_amd.elbo import loss_function  # noqa: E402

dev = torch.device('cuda:0')


def timeit(name, fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print('%-58s %8.1f us/step' % (name, (time.perf_counter() - t0) / n * 1e6), flush=True)


for B in (1024, 65536):
    L, K = 128, 256
    logits = torch.randn(B, 2 * L + K, device=dev, requires_grad=True)
    gz = torch.randn(B, L + K, device=dev) / B
    gkl = torch.full((B,), 1.0 / B, device=dev)
    m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'],
                              {'continuous_size': 2 * L, 'discrete_size': K, 'cuda': True, 'half': False}).train()

    def concat_step():
        z, params = m(logits)
        kl = m.kl(params)
        torch.autograd.backward([z, kl], [gz, gkl])
        logits.grad = None
    timeit('Concat(gauss 128, gumbel 256) fwd+kl+bwd  B=%d' % B, concat_step)

    g = IsotropicGaussian({'continuous_size': 2 * L, 'cuda': True, 'half': False}).train()
    lg = torch.randn(B, 2 * L, device=dev, requires_grad=True)
    x = (torch.rand(B, 1, 28, 28, device=dev) < 0.13).float()
    rc = torch.randn(B, 1, 28, 28, device=dev, requires_grad=True)
    gzz = torch.randn(B, L, device=dev) / B

    def elbo_step():
        z, params = g(lg)
        out = loss_function(rc, x, g.kl(params), 'bernoulli')
        torch.autograd.backward([out['loss_mean'], z], [None, gzz])
        lg.grad = None
        rc.grad = None
    timeit('IsotropicGaussian + loss_function fwd+bwd   B=%d' % B, elbo_step)
