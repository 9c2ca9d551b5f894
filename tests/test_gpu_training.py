"""Several optimizer steps of a small VAE whose latent/loss half is the HIP path, against the same model whose
latent/loss half is the oracle on the CPU: same initial weights, same injected noise every step.  Checks the
autograd plumbing end to end (stock encoder -> reparameterizer -> stock decoder -> loss_function -> backward ->
SGD) over many steps, the loss trajectory and the final weights.

What SimpleVAE / ParallellyReparameterizedVAE do around the path (simple_vae.py, parallelly_reparameterized_vae.py:
encode -> reparameterize -> decode -> loss_function(recon_x, x, params)) is restated with nn.Linear networks."""
import copy

import pytest
import torch
import torch.nn as nn

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class TinyVAE(nn.Module):
    """The Gaussian scale half goes through softplus + 0.5: the reference uses the raw encoder output as the scale
    (no exp), so KL has a -log|s| pole at s = 0 where a 1e-7 difference between the CPU and GPU GEMMs
    would be amplified without bound; away from the pole the two runs stay together to rounding."""

    def __init__(self, d_in, d_lat_in, d_lat_out, L):
        super().__init__()
        self.L = L
        self.enc = nn.Sequential(nn.Linear(d_in, 64), nn.Tanh(), nn.Linear(64, d_lat_in))
        self.dec = nn.Sequential(nn.Linear(d_lat_out, 64), nn.Tanh(), nn.Linear(64, d_in))

    def encode(self, x):
        h = self.enc(x.flatten(1))
        L = self.L
        return torch.cat([h[:, :L], torch.nn.functional.softplus(h[:, L:2 * L]) + 0.5, h[:, 2 * L:]], -1)


def run_hip(model, rep, xs, noises, lr, kl_beta):
    from vae_amd.elbo import ELBOMixin

    class Host(ELBOMixin):
        training = True
        config = {'nll_type': 'bernoulli', 'kl_beta': kl_beta, 'half': False, 'continuous_mut_info': 0,
                  'discrete_mut_info': 0}
        reparameterizer = rep
    host = Host()
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    losses = []
    for x, n in zip(xs, noises):
        xd = x.to(DEV)
        logits = model.encode(xd)
        z, params = rep(logits, noise=[t.to(DEV) for t in n] if isinstance(n, list) else n.to(DEV))
        recon = model.dec(z).view_as(xd)
        out = host.loss_function(recon, xd, params)
        opt.zero_grad()
        out['loss_mean'].backward()
        opt.step()
        losses.append(float(out['loss_mean'].detach()))
    return losses


def run_oracle(model, kind, xs, noises, lr, kl_beta, L, K):
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    losses = []
    for x, n in zip(xs, noises):
        logits = model.encode(x)
        if kind == 'gauss':
            z, params = O.gauss_forward(logits, n)
            kl = O.gauss_kl(params)
        else:
            z, params = O.concat_forward(logits, [('isotropic_gaussian', 2 * L, {}), ('discrete', K, {'tau': 1.0})], n)
            kl = O.concat_kl(params, ['isotropic_gaussian', 'discrete'])
        recon = model.dec(z).view_as(x)
        out = O.loss_function(O.nll(x, recon, 'bernoulli'), kl, kl_beta)
        opt.zero_grad()
        out['loss_mean'].backward()
        opt.step()
        losses.append(float(out['loss_mean'].detach()))
    return losses


@pytest.mark.parametrize('kind', ['gauss', 'concat'])
def test_training_trajectory_matches_oracle(kind):
    from vae_amd.reparameterizers import IsotropicGaussian, ConcatReparameterizer
    torch.manual_seed(0)
    B, L, K, steps = 128, 16, 10, 25
    d_lat_in, d_lat_out = (2 * L, L) if kind == 'gauss' else (2 * L + K, L + K)
    cpu_model = TinyVAE(784, d_lat_in, d_lat_out, L)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    cfg = {'continuous_size': 2 * L, 'discrete_size': K, 'cuda': True, 'half': False}
    rep = IsotropicGaussian(cfg) if kind == 'gauss' else ConcatReparameterizer(['isotropic_gaussian', 'discrete'], cfg)
    rep.train()
    xs = [(torch.rand(B, 1, 28, 28) < 0.13).float() for _ in range(steps)]
    if kind == 'gauss':
        noises = [torch.randn(B, L) for _ in range(steps)]
    else:
        noises = [[torch.randn(B, L), torch.rand(B, K)] for _ in range(steps)]
    l1 = run_hip(gpu_model, rep, xs, noises, lr=1e-3, kl_beta=0.5)
    l0 = run_oracle(cpu_model, kind, xs, noises, lr=1e-3, kl_beta=0.5, L=L, K=K)
    assert l0[-1] < l0[0]                                  # it trains
    torch.testing.assert_close(torch.tensor(l1), torch.tensor(l0), rtol=2e-5, atol=1e-3)
    for (n1, p1), (n0, p0) in zip(gpu_model.named_parameters(), cpu_model.named_parameters()):
        torch.testing.assert_close(p1.detach().cpu(), p0.detach(), rtol=1e-4, atol=1e-5, msg=lambda m: n1 + ': ' + m)
