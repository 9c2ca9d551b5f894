"""Tuning aid (GPU box): the Bernoulli NLL kernels at row lengths around 784 pixels with the same total bytes --
how much the ragged 3136-byte rows (24.5 cache lines) cost against 128-byte-aligned row lengths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import _lib
from vae_amd._lib import call, ptr
dev = torch.device('cuda', 0)
ws = _lib.workspace(dev); st = _lib.stream_ptr(dev)
for D in (768, 784, 800, 1024, 512):
    B = 65536 * 784 // D
    x = (torch.rand(B, D, device=dev) < 0.13).float(); recon = torch.randn(B, D, device=dev)
    kl = torch.rand(B, device=dev); nll = torch.empty(B, device=dev); loss = torch.empty(B, device=dev); elbo = torch.empty(B, device=dev)
    scal = torch.zeros(8, device=dev); grecon = torch.empty(B, D, device=dev); gm = torch.zeros(6, device=dev); gm[0] = 1
    gk = torch.empty(B, device=dev)
    fns = {
     'fused': (lambda: call('vae_nll_elbo_fused', ptr(x), ptr(recon), 0, ptr(kl), None, None, 1.0, 1.0 / B, 1.0, ptr(nll), ptr(loss), ptr(elbo), ptr(scal), ptr(ws), ptr(grecon), B, D, 0, st), (12 * D + 12) * B),
     'fwd': (lambda: call('vae_nll_elbo_fwd', ptr(x), ptr(recon), 0, ptr(kl), None, None, 1.0, 1.0 / B, ptr(nll), ptr(loss), ptr(elbo), ptr(scal), ptr(ws), B, D, 0, st), (8 * D + 12) * B),
     'bwd': (lambda: call('vae_nll_elbo_bwd', ptr(x), ptr(recon), 0, None, None, None, ptr(gm), 1.0, 1.0 / B, ptr(grecon), ptr(gk), None, None, B, D, 0, st), 12 * D * B),
    }
    for name, (fn, nbytes) in fns.items():
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('D=%4d B=%6d %-6s %.4f ms %.0f GB/s' % (D, B, name, ms, nbytes / ms / 1e6), flush=True)
