"""Where the host time of the eager C5 sequence goes (cProfile, small batch so the GPU never limits)."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import vrnn as V
from vae_amd.reparameterizers import IsotropicGaussian
dev = 'cuda:0'
Tn, B, L = 64, 256, 128
cfg = {'reparam_type': 'isotropic_gaussian', 'continuous_size': 2 * L, 'nll_type': 'bernoulli', 'kl_beta': 1.0,
       'use_prior_kl': True, 'cuda': True, 'half': False, 'continuous_mut_info': 0, 'discrete_mut_info': 0}
rep = IsotropicGaussian(cfg).train()
xs = [(torch.rand(B, 1, 28, 28, device=dev) < 0.3).float() for _ in range(Tn)]
gz = torch.randn(B, L, device=dev)
enc = [torch.randn(B, 2 * L, device=dev, requires_grad=True) for _ in range(Tn)]
pri = [torch.randn(B, 2 * L, device=dev, requires_grad=True) for _ in range(Tn)]
recons = [torch.randn(B, 1, 28, 28, device=dev, requires_grad=True) for _ in range(Tn)]


def fn():
    params, zs = [], []
    for t in range(Tn):
        z, p = V.reparameterize(rep, {'encoder_logits': enc[t], 'prior_logits': pri[t], 'x_features': None}, cfg)
        params.append(p)
        zs.append(z['posterior'])
    out = V.loss_function(rep, cfg, recons, xs, params)
    torch.autograd.backward([out['loss_mean']] + zs, [None] + [gz] * Tn, inputs=enc + pri + recons)
    for t in enc + pri + recons:
        t.grad = None


for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    fn()
torch.cuda.synchronize()
print('eager sequence: %.3f ms' % ((time.perf_counter() - t0) * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    fn()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
