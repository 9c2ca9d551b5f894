"""vae_amd.vrnn.loss_function over T = 64 steps of B = 1024 (BASELINE config 5), forward and backward, a few times:
the command behind the PMC traffic figure for the whole call (profiles/r02_vrnn_loss_traffic.txt):
    rocprofv3 --pmc FETCH_SIZE -- python3 scripts/vrnn_loss_call.py ; the same with WRITE_SIZE"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import vrnn as V
from vae_amd.reparameterizers import IsotropicGaussian

dev = torch.device('cuda', 0)
Tn, B, D, L = 64, 1024, 784, 128
cfg = {'reparam_type': 'isotropic_gaussian', 'continuous_size': 2 * L, 'nll_type': 'bernoulli', 'kl_beta': 1.0,
       'use_prior_kl': True, 'cuda': True, 'half': False}
rep = IsotropicGaussian(cfg).to(dev).train()
torch.manual_seed(0)
xs = [(torch.rand(B, 1, 28, 28, device=dev) < 0.3).float() for _ in range(Tn)]
recons = [torch.randn(B, 1, 28, 28, device=dev, requires_grad=True) for _ in range(Tn)]
params = []
for t in range(Tn):
    lm = {'encoder_logits': torch.randn(B, 2 * L, device=dev), 'prior_logits': torch.randn(B, 2 * L, device=dev),
          'x_features': None}
    params.append(V.reparameterize(rep, lm, cfg)[1])
for it in range(4):
    out = V.loss_function(rep, cfg, recons, xs, params)
    out['loss_mean'].backward()
torch.cuda.synchronize()
print('loss_mean', float(out['loss_mean']), 'algorithmic bytes per call: fwd', Tn * B * (8 * D + 16), 'bwd', Tn * B * 12 * D)
