"""Tuning aid (GPU box): interleaved A/B of library builds x runtime knobs in ONE process.

    python scripts/ab.py 'lib=a.so,b.so' 'G=auto,32' 'U=auto,4' 'MAXGRID=2048,1000000' [kernels=...]
Each variant is timed in rounds (5 launches per round, 12 rounds, interleaved); median per variant."""
import ctypes
import itertools
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import _lib

opts = dict(a.split('=', 1) for a in sys.argv[1:])
libs = opts.get('lib', 'vae_amd/libvae_amd.so').split(',')
knobs = {k: opts.get(k, 'auto').split(',') for k in ('G', 'U', 'MAXGRID')}
only = opts.get('kernels', '').split(',') if opts.get('kernels') else None
B, L, D = int(opts.get('B', 65536)), int(opts.get('L', 512)), int(opts.get('D', 784))
dev = torch.device('cuda', 0)
DT = torch.bfloat16 if opts.get('dtype') == 'bf16' else torch.float32
DC = 1 if DT == torch.bfloat16 else 0
ES = 2 if DC else 4
f32 = dict(dtype=DT, device=dev)
R = lambda *sh: torch.randn(*sh, device=dev).to(DT)
logits = R(B, 2 * L); noise = R(B, L)
x = (torch.rand(B, D, device=dev) < 0.13).to(DT); recon = R(B, D)
gz = R(B, L)
z = torch.empty(B, L, **f32); logvar = torch.empty(B, L, **f32); kl = torch.empty(B, **f32)
nll = torch.empty(B, **f32); loss = torch.empty(B, **f32); elbo = torch.empty(B, **f32)
scal = torch.zeros(8, device=dev); gmeans = torch.zeros(6, device=dev); gmeans[0] = 1
grecon = torch.empty(B, D, **f32); gkld = torch.empty(B, **f32); glogits = torch.empty(B, 2 * L, **f32)
ws = torch.zeros(_lib.load().vae_workspace_bytes() // 4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())

handles = {}
for path in libs:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argt in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).argtypes = argt
    handles[path] = lib


def kernels(lib):
    return {
        'gauss_fwd': (lambda: lib.vae_gauss_reparam_kl_fwd(p(logits), 2 * L, p(noise), p(z), 0, p(logvar), p(kl), p(scal), p(ws), B, L, 1e-6, 1.0 / (B * L), 1, 0, 0, 0, DC, st), (4 * L * ES + ES) * B),
        'gauss_fwd_nomeans': (lambda: lib.vae_gauss_reparam_kl_fwd(p(logits), 2 * L, p(noise), p(z), 0, p(logvar), p(kl), None, p(ws), B, L, 1e-6, 1.0 / (B * L), 1, 0, 0, 0, DC, st), (4 * L * ES + ES) * B),
        'gauss_fwd_nologvar': (lambda: lib.vae_gauss_reparam_kl_fwd(p(logits), 2 * L, p(noise), p(z), 0, None, p(kl), p(scal), p(ws), B, L, 1e-6, 1.0 / (B * L), 1, 0, 0, 0, DC, st), (4 * L * ES + ES) * B),
        'gauss_bwd': (lambda: lib.vae_gauss_reparam_kl_bwd(p(logits), 2 * L, p(noise), p(gz), 0, None, p(gkld), None, p(glogits), 0, B, L, 1e-6, 1.0 / (B * L), 1, 0, 0, 0, DC, st), (6 * L * ES + ES) * B),
        'nll_fwd': (lambda: lib.vae_nll_elbo_fwd(p(x), p(recon), 0, p(kl), None, None, 1.0, 1.0 / B, p(nll), p(loss), p(elbo), p(scal), p(ws), B, D, DC, st), (2 * D + 3) * ES * B),
        'nll_fused': (lambda: lib.vae_nll_elbo_fused(p(x), p(recon), 0, p(kl), None, None, 1.0, 1.0 / B, 1.0, p(nll), p(loss), p(elbo), p(scal), p(ws), p(grecon), B, D, DC, st), (3 * D + 4) * ES * B),
        'nll_bwd': (lambda: lib.vae_nll_elbo_bwd(p(x), p(recon), 0, None, None, None, p(gmeans), 1.0, 1.0 / B, p(grecon), p(gkld), None, None, B, D, DC, st), 3 * D * ES * B),
    }


variants = list(itertools.product(libs, knobs['G'], knobs['U'], knobs['MAXGRID']))


def setenv(G, U, MG):
    """The knobs are read from the environment once per library; in-process changes go through vae_tuning_set."""
    for lib in handles.values():
        for k, v in ((b'G', G), (b'U', U), (b'MAXGRID', MG)):
            lib.vae_tuning_set(k, 0 if v == 'auto' else int(v))


names = [n for n in kernels(handles[libs[0]]) if not only or n in only]
for kname in names:
    times = {v: [] for v in variants}
    for rnd in range(12):
        for v in variants:
            setenv(*v[1:])
            fn, nbytes = kernels(handles[v[0]])[kname]
            if rnd == 0:
                for _ in range(3):
                    assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5)
    for v in variants:
        med, mn = statistics.median(times[v]), min(times[v])
        print('%-10s %-28s G=%-4s U=%-4s grid=%-8s med %.4f ms %5.0f GB/s (%.1f%%)  min %.4f ms %5.0f GB/s' % (
            kname, os.path.basename(v[0]), v[1], v[2], v[3], med, nbytes / med / 1e6, nbytes / med / 1e6 / 80,
            mn, nbytes / mn / 1e6), flush=True)
