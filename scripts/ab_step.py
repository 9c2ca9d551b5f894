"""Tuning aid (GPU box): the bench.py step sequence (C2: gauss fwd -> one-pass NLL -> loss backward -> gauss bwd)
under several settings of the launch-geometry knobs, interleaved in ONE process (vae_tuning_set); VAE_AMD_LIB selects
a variant build of the library (scripts/build_variant.sh) for the whole run.

    python scripts/ab_step.py [variants=U:G:MAXGRID:REDUCEGRID,...] [B=65536] [L=512] [D=784] [rounds=6] [steps=30]

Per variant: median whole-step time (no events inside the loop) and, from a second pass with HIP events around each
launch, the median duration of the two reducing calls (gauss fwd, one-pass NLL; the closing launch of RMODE 2 is
inside the bracket).  Also checks the eight batch means of every variant against the first one."""
import ctypes
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import _lib
from vae_amd._lib import call, ptr

opts = dict(a.split('=', 1) for a in sys.argv[1:])
B, L, D = int(opts.get('B', 65536)), int(opts.get('L', 512)), int(opts.get('D', 784))
rounds, steps = int(opts.get('rounds', 6)), int(opts.get('steps', 30))
DT = {'bf16': torch.bfloat16, 'f16': torch.float16}.get(opts.get('dtype'), torch.float32)
DC = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[DT]
variants = [tuple(int(v) for v in s.split(':')) for s in opts.get(
    'variants', '0:0:0').split(',')]
dev = torch.device('cuda', 0)
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(1234)
mk = lambda *sh: torch.randn(*sh, generator=g, device=dev).to(DT)
logits, noise, recon = mk(B, 2 * L), mk(B, L), mk(B, D)
x = (torch.rand(B, D, generator=g, device=dev) < 0.13).to(DT)
gz = (torch.randn(B, L, generator=g, device=dev) / B).to(DT)
E = lambda *sh: torch.empty(*sh, dtype=DT, device=dev)
z, kl, nll, loss, elbo, grecon, gkld, glogits = E(B, L), E(B), E(B), E(B), E(B), E(B, D), E(B), E(B, 2 * L)
scal = torch.zeros(8, device=dev)
gmeans = torch.zeros(6, device=dev)
gmeans[0] = 1
ws, st = _lib.workspace(dev), _lib.stream_ptr(dev)
pm = ctypes.c_void_p(scal.data_ptr() + 24)
inv, ms = 1.0 / B, 1.0 / (B * L)


def setv(v):
    for n, val in zip((b'U', b'G', b'MAXGRID', b'REDUCEGRID'), tuple(v) + (0,) * (4 - len(v))):
        assert lib.vae_tuning_set(n, val) == 0


def step(ev=None):
    def mark(i):
        if ev is not None:
            ev[i].record()
    mark(0)
    call('vae_gauss_reparam_kl_fwd', ptr(logits), 2 * L, ptr(noise), ptr(z), L, None, ptr(kl), pm, ptr(ws), B, L, 1e-6,
         ms, 1, 0, 0, 0, DC, st)
    mark(1)
    call('vae_nll_elbo_fused', ptr(x), ptr(recon), 0, ptr(kl), None, None, 1.0, inv, 1.0, ptr(nll), ptr(loss),
         ptr(elbo), ptr(scal), ptr(ws), ptr(grecon), B, D, DC, st)
    mark(2)
    call('vae_nll_fused_backward', ptr(grecon), B * D, ptr(gmeans), 1.0, inv, ptr(gkld), None, None, B, DC, st)
    mark(3)
    call('vae_gauss_reparam_kl_bwd', ptr(logits), 2 * L, ptr(noise), ptr(gz), L, None, ptr(gkld), None, ptr(glogits),
         2 * L, B, L, 1e-6, ms, 1, 0, 0, 0, DC, st)
    mark(4)


ref = None
for v in variants:
    setv(v)
    scal.zero_()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    got = scal.clone()
    if ref is None:
        ref = got
    else:
        torch.testing.assert_close(got, ref, rtol=2e-6, atol=1e-7, msg=lambda m: 'variant %s: %s' % (v, m))
    # determinism: the same variant twice gives the same bits
    step()
    torch.cuda.synchronize()
    assert torch.equal(scal, got), 'variant %s not deterministic' % (v,)
if ref is not None:
    print('means agree across %d variants: %s' % (len(variants), [round(float(t), 6) for t in ref]), flush=True)

whole = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        setv(v)
        step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        whole[v].append(e0.elapsed_time(e1) / steps)
per = {v: [[], [], [], []] for v in variants}
for r in range(rounds):
    for v in variants:
        setv(v)
        step()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(10)]
        for ev in evs:
            step(ev)
        torch.cuda.synchronize()
        for k in range(4):
            per[v][k].append(statistics.median(ev[k].elapsed_time(ev[k + 1]) for ev in evs))
ES = 2 if DC else 4
ab = [(3 * L * ES + ES) * B + L * ES * B, (3 * D * ES + 3 * ES) * B, 0, (5 * L * ES + ES) * B + L * ES * B]
for v in variants:
    m = [statistics.median(p) * 1e3 for p in per[v]]
    w = statistics.median(whole[v]) * 1e3
    print('%-14s step %7.1f us (min %7.1f) %6.1f M samples/s | gauss_fwd %6.1f us %.3f | '
          'nll_fused %6.1f us %.3f | lossbwd %5.1f | gauss_bwd %6.1f us %.3f' % (
              (':'.join(map(str, v)), w, min(whole[v]) * 1e3, B / w, m[0], ab[0] / m[0] / 8e6, m[1], ab[1] / m[1] / 8e6, m[2], m[3],
                   ab[3] / m[3] / 8e6)), flush=True)
