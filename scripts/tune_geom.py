"""Tuning aid (GPU box): time each hot-path kernel of the bench workload under VAE_AMD_G overrides."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import _lib
from vae_amd._lib import call, ptr

B, L, D = 65536, 512, 784
dev = torch.device('cuda', 0)
f32 = dict(dtype=torch.float32, device=dev)
logits = torch.randn(B, 2 * L, **f32); noise = torch.randn(B, L, **f32)
x = (torch.rand(B, D, **f32) < 0.13).float(); recon = torch.randn(B, D, **f32)
gz = torch.randn(B, L, **f32)
z = torch.empty(B, L, **f32); logvar = torch.empty(B, L, **f32); kl = torch.empty(B, **f32)
nll = torch.empty(B, **f32); loss = torch.empty(B, **f32); elbo = torch.empty(B, **f32)
scal = torch.zeros(8, **f32); gmeans = torch.zeros(6, **f32); gmeans[0] = 1
grecon = torch.empty(B, D, **f32); gkld = torch.empty(B, **f32); glogits = torch.empty(B, 2 * L, **f32)
ws = _lib.workspace(dev); st = _lib.stream_ptr(dev)
ks = {
    'gauss_fwd': (lambda: call('vae_gauss_reparam_kl_fwd', ptr(logits), 2 * L, ptr(noise), ptr(z), 0, ptr(logvar), ptr(kl), ptr(scal), ptr(ws), B, L, 1e-6, 1.0 / (B * L), 1, 0, 0, 0, 0, st), (16 * L + 4) * B),
    'gauss_fwd_philox': (lambda: call('vae_gauss_reparam_kl_fwd', ptr(logits), 2 * L, None, ptr(z), 0, ptr(logvar), ptr(kl), ptr(scal), ptr(ws), B, L, 1e-6, 1.0 / (B * L), 1, 1, 1, 0, 0, st), (12 * L + 4) * B),
    'gauss_bwd': (lambda: call('vae_gauss_reparam_kl_bwd', ptr(logits), 2 * L, ptr(noise), ptr(gz), 0, None, ptr(gkld), None, ptr(glogits), 0, B, L, 1e-6, 1.0 / (B * L), 1, 0, 0, 0, 0, st), (24 * L + 4) * B),
    'nll_fwd': (lambda: call('vae_nll_elbo_fwd', ptr(x), ptr(recon), 0, ptr(kl), None, None, 1.0, 1.0 / B, ptr(nll), ptr(loss), ptr(elbo), ptr(scal), ptr(ws), B, D, 0, st), (8 * D + 12) * B),
    'nll_fused': (lambda: call('vae_nll_elbo_fused', ptr(x), ptr(recon), 0, ptr(kl), None, None, 1.0, 1.0 / B, 1.0, ptr(nll), ptr(loss), ptr(elbo), ptr(scal), ptr(ws), ptr(grecon), B, D, 0, st), (12 * D + 12) * B),
    'nll_bwd': (lambda: call('vae_nll_elbo_bwd', ptr(x), ptr(recon), 0, None, None, None, ptr(gmeans), 1.0, 1.0 / B, ptr(grecon), ptr(gkld), None, None, B, D, 0, st), 12 * D * B),
}
only = os.environ.get('TUNE_ONLY')
for name, (fn, nbytes) in ks.items():
    if only and name not in only.split(','):
        continue
    for spec in (sys.argv[1:] or ['']):   # 'auto' | 'G' | 'GxU'
        G, _, Uo = ('' if spec == 'auto' else spec).partition('x')
        for key, val in (('VAE_AMD_G', G), ('VAE_AMD_U', Uo)):
            if val:
                os.environ[key] = val
            else:
                os.environ.pop(key, None)
        G = spec
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%-18s G=%-4s %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)' % (name, G or 'auto', ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80), flush=True)
