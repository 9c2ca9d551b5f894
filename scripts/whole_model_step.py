#!/usr/bin/env python3
"""SURVEY 8(d) 'ELBO-step samples/sec = full fwd+bwd of (stock encoder/decoder + our kernels)': the example's
SimpleVAE (nn.Linear 784-512-2L / L-512-784 networks, Adam) at configs[1]'s shape, once with the drop-in latent +
loss half and once with the reference's unfused torch op sequence in the same place (isotropic_gaussian.py:77-131,
158-170 through torch.distributions; BCE-with-logits NLL; abstract_vae.py:356-387).  Same weights, same data."""
import argparse
import os
import sys
import time

import torch
import torch.distributions as D
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
from train_simple_vae import SimpleVAE  # noqa: E402


def stock_step(model, x, L):
    """The reference's latent + loss ops in stock PyTorch around the same encoder / decoder."""
    logits = model.encoder(x.flatten(1))
    mu, s = logits[:, :L], logits[:, L:] + 1e-6
    z = torch.zeros_like(s).normal_() * (0.5 * s) + mu
    recon = model.decoder(z).view_as(x)
    kld = D.kl_divergence(D.Normal(mu, s), D.Normal(torch.zeros_like(mu), torch.ones_like(s))).sum(-1)
    nll = F.binary_cross_entropy_with_logits(recon, x, reduction='none').flatten(1).sum(-1)
    loss = nll + model.config['kl_beta'] * kld
    return {'loss_mean': loss.mean(), 'elbo_mean': (nll + kld).mean(), 'nll_mean': nll.mean(), 'kld_mean': kld.mean(),
            'mu_mean': mu.mean(), 'logvar_mean': s.mean()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=65536)
    ap.add_argument('--latent', type=int, default=512)
    ap.add_argument('--steps', type=int, default=20)
    args = ap.parse_args()
    D.Distribution.set_default_validate_args(False)
    dev = torch.device('cuda:0')
    B, L = args.batch, args.latent
    config = {'reparam_type': 'isotropic_gaussian', 'continuous_size': 2 * L, 'nll_type': 'bernoulli', 'kl_beta': 1.0,
              'cuda': True, 'half': False, 'continuous_mut_info': 0, 'discrete_mut_info': 0}
    x = (torch.rand(B, 1, 28, 28, device=dev) < 0.13).float()
    for name in ('vae_amd', 'stock torch'):
        torch.manual_seed(0)
        model = SimpleVAE(config).to(dev).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)

        def step():
            if name == 'vae_amd':
                recon, params = model(x)
                loss = model.loss_function(recon, x, params)
            else:
                loss = stock_step(model, x, L)
            opt.zero_grad(set_to_none=True)
            loss['loss_mean'].backward()
            opt.step()
            return loss['loss_mean']
        for _ in range(5):
            last = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print('%-12s B=%d L=%d  %.3f ms/step  %.2f M samples/s  (loss %.2f)' % (name, B, L, dt * 1e3, B / dt / 1e6,
                                                                              float(last)), flush=True)


if __name__ == '__main__':
    main()
