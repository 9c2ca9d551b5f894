"""Per-kernel roofline table for the BASELINE.json configs other than the bench.py headline (GPU box).

    python scripts/bench_configs.py [--json out.json]

Every kernel is called through the C ABI on preallocated buffers; bytes = algorithmic bytes of that
launch.  Time: launches that move >= 32 MB are timed ONE AT A TIME with HIP events, each behind a 512 MiB fill of
another buffer, median of 9 -- the caches are then as cold as the kernels of a training step find them (r02: looping
one kernel on its own working set let the 256 MiB Infinity Cache hold part of it, which flattered plain stores and
hid what non-temporal stores gain; --loop restores that form).  Smaller launches (latency-bound): HIP events around 20
back-to-back launches after 5 warm-ups.
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import _lib
from vae_amd._lib import call, ptr
from vae_amd.timing import Event

dev = torch.device('cuda', 0)
ws = _lib.workspace(dev)
st = _lib.stream_ptr(dev)
rows = []


def T(*shape, dtype=torch.float32, kind='randn'):
    if kind == 'randn':
        return torch.randn(*shape, device=dev, dtype=torch.float32).to(dtype)
    if kind == 'rand':
        return torch.rand(*shape, device=dev, dtype=torch.float32).to(dtype)
    return torch.empty(*shape, device=dev, dtype=dtype)


LOOP = '--loop' in sys.argv
QUIET = False      # bench.py imports the config functions and wants their rows, not their lines
_flush = None


def timeit(name, cfg, fn, nbytes, iters=20):
    global _flush
    for _ in range(5):
        fn()
    # events without a system-scope fence (vae_amd/timing.py): ~2 us closer to rocprofv3's kernel durations
    e0, e1 = Event(), Event()
    torch.cuda.synchronize()
    if nbytes >= (32 << 20) and not LOOP:
        if _flush is None:
            _flush = torch.empty(512 << 18, device=dev)      # 512 MiB of fp32
        samples = []
        for _ in range(int(os.environ.get('BENCH_COLD_SAMPLES', '9'))):
            _flush.fill_(1.0)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1))
        ms = sorted(samples)[len(samples) // 2]
    else:
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
    gbs = nbytes / ms / 1e6
    rows.append({'kernel': name, 'config': cfg, 'ms': round(ms, 5), 'algorithmic_MB': round(nbytes / 1e6, 1),
                 'GBs': round(gbs), 'frac_of_8TBs': round(gbs / 8000, 3)})
    if not QUIET:
        print('%-26s %-44s %9.4f ms %8.1f MB %6.0f GB/s  %5.1f%%' % (name, cfg, ms, nbytes / 1e6, gbs, gbs / 80), flush=True)


def gauss(B, L, dtype=torch.float32, philox=False):
    es = 2 if dtype == torch.bfloat16 else 4
    dc = 1 if dtype == torch.bfloat16 else 0
    logits, noise, gz = T(B, 2 * L, dtype=dtype), T(B, L, dtype=dtype), T(B, L, dtype=dtype)
    z, lv, kl, gkl = T(B, L, dtype=dtype, kind='e'), T(B, L, dtype=dtype, kind='e'), T(B, dtype=dtype, kind='e'), T(B, dtype=dtype)
    gl, means = T(B, 2 * L, dtype=dtype, kind='e'), torch.zeros(2, device=dev)
    n = None if philox else noise
    cfg = 'B=%d L=%d %s%s' % (B, L, 'bf16' if dc else 'f32', ' philox' if philox else '')
    # the training-step form: logvar is produced on first read (vae_amd/lazy.py), the launch writes z and kl
    timeit('gauss_reparam_kl_fwd', cfg, lambda: call('vae_gauss_reparam_kl_fwd', ptr(logits), 2 * L, ptr(n), ptr(z), 0, None, ptr(kl), ptr(means), ptr(ws), B, L, 1e-6, 1.0 / (B * L), 1, 1, 1, 0, dc, st),
           B * ((3 if philox else 4) * L * es + es))
    timeit('gauss_reparam_kl_fwd', cfg + ' +logvar', lambda: call('vae_gauss_reparam_kl_fwd', ptr(logits), 2 * L, ptr(n), ptr(z), 0, ptr(lv), ptr(kl), ptr(means), ptr(ws), B, L, 1e-6, 1.0 / (B * L), 1, 1, 1, 0, dc, st),
           B * ((4 if philox else 5) * L * es + es))
    timeit('gauss_reparam_kl_bwd', cfg, lambda: call('vae_gauss_reparam_kl_bwd', ptr(logits), 2 * L, ptr(n), ptr(gz), 0, None, ptr(gkl), None, ptr(gl), 0, B, L, 1e-6, 1.0 / (B * L), 1, 1, 1, 0, dc, st),
           B * ((5 if philox else 6) * L * es + es))


def gauss_pair(B, L):
    mq, sq, mp, sp = T(B, L), T(B, L), T(B, L), T(B, L, kind='rand') + 0.1
    kl, gkl = T(B, kind='e'), T(B)
    g = [T(B, L, kind='e') for _ in range(4)]
    cfg = 'B=%d L=%d f32 (VRNN post||prior)' % (B, L)
    timeit('gauss_kl_fwd', cfg, lambda: call('vae_gauss_kl_fwd', ptr(mq), L, ptr(sq), L, ptr(mp), L, ptr(sp), L, ptr(kl), B, L, 0, st), B * (16 * L + 4))
    timeit('gauss_kl_bwd', cfg, lambda: call('vae_gauss_kl_bwd', ptr(mq), L, ptr(sq), L, ptr(mp), L, ptr(sp), L, ptr(gkl), ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]), B, L, 0, st), B * (32 * L + 4))


def pair_fused(B, L):
    enc, pri, n1, n2 = T(B, 2 * L), T(B, 2 * L), T(B, L), T(B, L)
    zq, zp, lvq, lvp, cl = T(B, L, kind='e'), T(B, L, kind='e'), T(B, L, kind='e'), T(B, L, kind='e'), T(B, 2 * L, kind='e')
    klp, kl0, gk = T(B, kind='e'), T(B, kind='e'), T(B)
    means = torch.zeros(4, device=dev)
    ge, gp = T(B, 2 * L, kind='e'), T(B, 2 * L, kind='e')
    cfg = 'B=%d L=%d f32 (one VRNN step, fused)' % (B, L)
    timeit('gauss_pair_fwd', cfg, lambda: call('vae_gauss_pair_fwd', ptr(enc), 2 * L, ptr(pri), 2 * L, ptr(n1), ptr(n2), ptr(zq), ptr(zp), ptr(lvq), ptr(lvp), ptr(cl), ptr(klp), ptr(kl0), ptr(means), ptr(ws), B, L, 1e-6, 1.0 / (B * L), 1, 0, 0, 0, 0, st),
           B * (40 * L + 4))   # SURVEY: 2 logits 16L + 2 noise 8L + 2 z 8L + clamped prior logits 8L + kl
    timeit('gauss_pair_fwd (no logvars)', cfg, lambda: call('vae_gauss_pair_fwd', ptr(enc), 2 * L, ptr(pri), 2 * L, ptr(n1), ptr(n2), ptr(zq), ptr(zp), None, None, ptr(cl), ptr(klp), ptr(kl0), ptr(means), ptr(ws), B, L, 1e-6, 1.0 / (B * L), 1, 0, 0, 0, 0, st),
           B * (40 * L + 4))
    timeit('gauss_pair_bwd', cfg, lambda: call('vae_gauss_pair_bwd', ptr(enc), 2 * L, ptr(pri), 2 * L, ptr(n1), ptr(n2), ptr(zq), ptr(zp), None, None, None, ptr(gk), ptr(gk), None, ptr(ge), ptr(gp), B, L, 1e-6, 1.0 / (B * L), 1, 0, 0, 0, 0, st),
           B * (48 * L + 8))   # 2 logits 16L + 2 noise 8L + 2 gz 8L + 2 glogits 16L


def gumbel(B, K, philox=False):
    logits, U = T(B, K), T(B, K, kind='rand')
    y, zh, lq, gy, gl = [T(B, K, kind='e') for _ in range(5)]
    gy.normal_()
    u = None if philox else U
    cfg = 'B=%d K=%d f32%s' % (B, K, ' philox' if philox else '')
    timeit('gumbel_fwd', cfg, lambda: call('vae_gumbel_fwd', ptr(logits), K, ptr(u), ptr(y), 0, ptr(zh), 0, ptr(lq), B, K, 0.7, 1, 1, 0, 0, st), B * ((16 if philox else 20) * K))
    timeit('gumbel_fwd (y only: training)', cfg, lambda: call('vae_gumbel_fwd', ptr(logits), K, ptr(u), ptr(y), 0, None, 0, None, B, K, 0.7, 1, 1, 0, 0, st), B * ((8 if philox else 12) * K))
    timeit('gumbel_fwd (log_q_z only)', cfg, lambda: call('vae_gumbel_fwd', ptr(logits), K, None, None, 0, None, 0, ptr(lq), B, K, 0.7, 1, 1, 0, 0, st), B * 8 * K)
    timeit('gumbel_bwd (gy only)', cfg, lambda: call('vae_gumbel_bwd', ptr(y), 0, ptr(lq), ptr(gy), 0, None, 0, None, ptr(gl), 0, B, K, 0.7, 0, st), B * 12 * K)
    kl, gkl = T(B, kind='e'), T(B)
    timeit('categorical_kl_fwd', cfg, lambda: call('vae_categorical_kl_fwd', ptr(lq), K, ptr(logits), K, ptr(kl), B, K, 0, st), B * (8 * K + 4))
    timeit('categorical_kl_bwd', cfg, lambda: call('vae_categorical_kl_bwd', ptr(lq), K, ptr(logits), K, ptr(gkl), ptr(y), ptr(zh), B, K, 0, st), B * (16 * K + 4))


def concat(B, L, K):
    """Concat(Gaussian L, Gumbel K) on one [B, 2L+K] logits buffer: the per-child launches over strided slices (r02)
    and the one-launch row kernel (csrc/concat.hip), eval (all outputs) and training (sample row only)."""
    W = 2 * L + K
    logits, noise, U = T(B, W), T(B, L), T(B, K, kind='rand')
    z, lv, kl, means = T(B, L, kind='e'), T(B, L, kind='e'), T(B, kind='e'), torch.zeros(2, device=dev)
    y, zh, lq = [T(B, K, kind='e') for _ in range(3)]
    zcat, gz, gkl, gl = T(B, L + K, kind='e'), T(B, L + K), T(B), T(B, W, kind='e')
    disc = ctypes.c_void_p(logits.data_ptr() + 2 * L * 4)
    shape = 'B=%d L=%d K=%d f32' % (B, L, K)
    ms = 1.0 / (B * L)

    def fn():
        call('vae_gauss_reparam_kl_fwd', ptr(logits), W, ptr(noise), ptr(z), 0, ptr(lv), ptr(kl), ptr(means), ptr(ws), B, L, 1e-6, ms, 1, 0, 0, 0, 0, st)
        call('vae_gumbel_fwd', disc, W, ptr(U), ptr(y), 0, ptr(zh), 0, ptr(lq), B, K, 1.0, 0, 0, 0, 0, st)
    timeit('concat fwd (gauss+gumbel)', shape + ', per-child launches on strided views, all outputs', fn, B * (16 * L + 4 + 20 * K))

    def row(training, inj):
        return lambda: call('vae_concat_gauss_gumbel_fwd', ptr(logits), W, ptr(noise) if inj and training else None, ptr(U) if inj else None,
                            ptr(zcat), 0, None if training else ptr(y), None if training else ptr(lq), ptr(kl),
                            B, L, K, 1e-6, 1.0, int(training), 1, 1, 1, 2, 0, 0, st)
    # eval, all outputs: logits 4W + noise-free Gaussian (z = mu) + U 4K in; [mu | z_hard] 4(L+K) + y 4K + log_q_z 4K + kl out
    timeit('concat fwd row kernel', shape + ' eval: [mu|z_hard] + y + log_q_z, injected U', row(False, True),
           B * (4 * W + 4 * K + 4 * (L + K) + 8 * K + 4))
    timeit('concat fwd row kernel', shape + ' training: [z|y], injected noise', row(True, True),
           B * (4 * W + 4 * (L + K) + 4 * (L + K) + 4))
    timeit('concat fwd row kernel', shape + ' training: [z|y], in-kernel Philox', row(True, False),
           B * (4 * W + 4 * (L + K) + 4))

    def bwd(inj):
        return lambda: call('vae_concat_gauss_gumbel_bwd', ptr(logits), W, ptr(noise) if inj else None, ptr(zcat), 0, ptr(gz), 0,
                            ptr(gkl), ptr(gl), 0, B, L, K, 1e-6, 1.0, 1, 1, 0, 0, st)
    # logits' Gaussian part 8L + noise 4L + gz 4(L+K) + y 4K + gkl in; d(logits) 4W out
    timeit('concat bwd row kernel', shape + ' training, injected noise', bwd(True), B * (8 * L + 4 * L + 4 * (L + K) + 4 * K + 4 + 4 * W))
    timeit('concat bwd row kernel', shape + ' training, in-kernel Philox', bwd(False), B * (8 * L + 4 * (L + K) + 4 * K + 4 + 4 * W))


def concat_module(B, L, K):
    """Whole ConcatReparameterizer step through the drop-in module (forward, kl, backward of z.w + kl):
    the one-node strided path against the per-child path (narrow views, torch.cat, narrow-backward adds)."""
    from vae_amd.reparameterizers import ConcatReparameterizer, concat_reparameterizer as C
    W = 2 * L + K
    logits = T(B, W).requires_grad_(True)
    noise, w = [T(B, L), T(B, K, kind='rand')], T(B, L + K)
    cfgd = {'continuous_size': 2 * L, 'discrete_size': K, 'cuda': True, 'half': False}
    # fwd: logits, noise, zcat, logvar, z_hard, log_q_z, kl   bwd: gauss logits, eps, gz, y, log_q_z, d(logits)
    nbytes = B * 4 * ((W + L + K + L + K + L + 2 * K + 1) + (2 * L + L + L + K + 2 * K + W))
    fusable = C._FUSABLE
    for label, fus in (('concat step (one node)', fusable), ('concat step (per child)', ())):
        C._FUSABLE = fus
        m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'], dict(cfgd)).train()

        def fn():
            z, params = m(logits, noise=noise)
            ((z * w).sum() + m.kl(params).sum()).backward()
            logits.grad = None
        timeit(label, 'B=%d L=%d K=%d f32 module fwd+kl+bwd (+ torch z*w sum)' % (B, L, K), fn, nbytes)
    C._FUSABLE = fusable
    # the same one-node step with in-kernel Philox draws, eager and replayed from one HIP graph (the step is
    # host-bound eager: ~135 us of Python and autograd around ~130 us of kernels)
    m = ConcatReparameterizer(['isotropic_gaussian', 'discrete'], dict(cfgd)).train()

    gkl = torch.ones(B, device=dev)

    def fn2():
        z, params = m(logits)
        torch.autograd.backward([z, m.kl(params)], [w, gkl])      # the decoder's d z and d(sum kl): no torch kernels
        logits.grad = None
    # the row kernels' bytes: fwd logits 4W + sample 4(L+K) + kl; bwd Gaussian logits 8L + gz 4(L+K) + y 4K + gkl + d(logits) 4W
    nb2 = B * ((4 * W + 4 * (L + K) + 4) + (8 * L + 4 * (L + K) + 4 * K + 4 + 4 * W))
    cfg2 = 'B=%d L=%d K=%d f32 module fwd+kl+bwd, in-kernel noise' % (B, L, K)
    timeit('concat step (eager)', cfg2, fn2, nb2)
    # a fresh leaf for the captured step: autograd ties a leaf's AccumulateGrad node to the stream of its first
    # backward; one created on the default stream drags that stream into the capture (torch warns, HIP then
    # crashes in hipStreamEndCapture)
    lg = logits.detach().clone().requires_grad_(True)

    def fn3():
        z, params = m(lg)
        torch.autograd.backward([z, m.kl(params)], [w, gkl])
        lg.grad = None
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn3()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn3()
    timeit('concat step (HIP graph)', cfg2, graph.replay, nb2)


def multi(B, L, D, S):
    """(f)2 multi-sample forms: S draws from one logits read; S decoder outputs against one read of x."""
    logits, noise, gz = T(B, 2 * L), T(S, B, L), T(S, B, L)
    z, lv, kl, gkl, gl = T(S, B, L, kind='e'), T(B, L, kind='e'), T(B, kind='e'), T(B), T(B, 2 * L, kind='e')
    means = torch.zeros(2, device=dev)
    cfg = 'B=%d L=%d S=%d f32' % (B, L, S)
    timeit('gauss_multi_fwd', cfg, lambda: call('vae_gauss_reparam_multi_fwd', ptr(logits), 2 * L, ptr(noise), ptr(z), ptr(lv), ptr(kl), ptr(means), ptr(ws), B, L, S, 1e-6, 1.0 / (B * L), 0, 0, 0, 0, st),
           B * (8 * L + 4 * L + S * 8 * L + 4))       # logits + logvar + S (noise + z) + kl
    timeit('gauss_multi_bwd', cfg, lambda: call('vae_gauss_reparam_multi_bwd', ptr(logits), 2 * L, ptr(noise), ptr(gz), None, ptr(gkl), None, ptr(gl), B, L, S, 1e-6, 1.0 / (B * L), 0, 0, 0, 0, st),
           B * (8 * L + S * 8 * L + 8 * L + 4))       # logits + S (noise + gz) + glogits
    x = (T(B, D, kind='rand') < 0.13).float()
    recons, grecons = [T(B, D) for _ in range(S)], [T(B, D, kind='e') for _ in range(S)]
    kld, n, lo, el = T(B, kind='rand'), T(B, kind='e'), T(B, kind='e'), T(B, kind='e')
    m6, gm = torch.zeros(6, device=dev), torch.zeros(6, device=dev)
    gm[0] = 1
    rp, gp = _lib.ptr_array(recons), _lib.ptr_array(grecons)
    cfg = 'B=%d D=%d S=%d bernoulli f32' % (B, D, S)
    timeit('nll_multi_elbo_fwd', cfg, lambda: call('vae_nll_multi_elbo_fwd', ptr(x), rp, S, 0, ptr(kld), None, None, 1.0, 1.0 / B, ptr(n), ptr(lo), ptr(el), ptr(m6), ptr(ws), B, D, 0, st),
           B * (4 * D + S * 4 * D + 12))
    timeit('nll_multi_elbo_bwd', cfg, lambda: call('vae_nll_multi_elbo_bwd', ptr(x), rp, gp, S, 0, None, None, None, ptr(gm), 1.0, 1.0 / B, None, None, None, B, D, 0, st),
           B * (4 * D + S * 8 * D))


def likelihood(B, K, L=64, D=784):
    """AbstractVAE.likelihood over one minibatch (abstract_vae.py:281-317): the batched form against the reference's
    loop (selected by an overridden reparameterize), wall time of the whole call; decoder = one nn.Linear."""
    import time
    import torch.nn as nn
    from vae_amd.elbo import ELBOMixin
    from vae_amd.reparameterizers import IsotropicGaussian

    def model(mixin):
        class M(mixin, nn.Module):
            def __init__(self):
                nn.Module.__init__(self)
                self.config = {'continuous_size': 2 * L, 'nll_type': 'bernoulli', 'kl_beta': 1.0, 'cuda': True,
                               'half': False, 'continuous_mut_info': 0, 'discrete_mut_info': 0,
                               'reparam_type': 'isotropic_gaussian'}
                self.reparameterizer = IsotropicGaussian(self.config)
                self.enc, self.dec = nn.Linear(D, 2 * L), nn.Linear(L, D)

            def encode(self, x):
                return self.enc(x)

            def decode(self, z):
                return self.dec(z)
        torch.manual_seed(0)
        m = M().to(dev).eval()
        m.reparameterizer.train(True)
        return m

    class Loop(ELBOMixin):
        def reparameterize(self, logits, labels=None, force=False):
            return self.reparameterizer(logits, force=force)

    x = (torch.rand(B, D, device=dev) < 0.3).float()
    loader = [(x, torch.zeros(B, dtype=torch.long, device=dev))]
    for name, mixin in (('likelihood (batched)', ELBOMixin), ('likelihood (reference loop)', Loop)):
        m = model(mixin)
        m.likelihood(loader, K=K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        val = float(m.likelihood(loader, K=K))
        ms = (time.perf_counter() - t0) * 1e3
        draws = B * (K // B) * B
        print('%-26s %-44s %9.2f ms  %d decoded draws, %.1f M draws/s, value %.3f' % (
            name, 'B=%d K=%d L=%d %dpx bernoulli f32' % (B, K, L, D), ms, draws, draws / ms / 1e3, val), flush=True)
        rows.append({'kernel': name, 'config': 'B=%d K=%d L=%d D=%d' % (B, K, L, D), 'ms': round(ms, 3), 'draws': draws})


def nll_steps(Tn, B, D):
    """The VRNN loss over Tn steps (vrnn.py:805-829) through the step-table entry points: Tn separate [B, D] target and
    reconstruction tensors read where they are (no torch.cat), and the same call through vae_amd.vrnn.loss_function."""
    xs = [(T(B, D, kind='rand') < 0.3).float() for _ in range(Tn)]
    rs = [T(B, D) for _ in range(Tn)]
    gs = [T(B, D, kind='e') for _ in range(Tn)]
    kld = T(Tn * B, kind='rand')
    nll_, loss, elbo = T(Tn * B, kind='e'), T(Tn * B, kind='e'), T(Tn * B, kind='e')
    la, ea, means = T(B, kind='e'), T(B, kind='e'), torch.zeros(6, device=dev)
    gmeans = torch.zeros(6, device=dev)
    gmeans[0] = 1
    gkld = T(Tn * B, kind='e')
    px, pr, pg = _lib.ptr_array(xs), _lib.ptr_array(rs), _lib.ptr_array(gs)
    cfg = 'T=%d x B=%d D=%d bernoulli f32, %d tensors each side' % (Tn, B, D, Tn)
    inv = 1.0 / (Tn * B)
    timeit('nll_steps_elbo_fwd', cfg, lambda: call('vae_nll_steps_elbo_fwd', px, pr, Tn, 0, ptr(kld), None, None, None, 1.0, inv, ptr(nll_),
                                                   ptr(loss), ptr(elbo), ptr(la), ptr(ea), None, ptr(means), ptr(ws), B, D,
                                                   0, st), Tn * B * (8 * D + 16))
    timeit('nll_steps_elbo_bwd', cfg, lambda: call('vae_nll_steps_elbo_bwd', px, pr, pg, Tn, 0, None, None, None,
                                                   ptr(gmeans), None, 1.0, inv, ptr(gkld), None, None, B, D, 0, st),
           Tn * B * (12 * D + 4))


def vrnn_sequence(Tn, B, L, D=784):
    """BASELINE config 5 end to end on the hot path: Tn fused posterior/prior steps (vae_amd.vrnn.reparameterize), the
    loss over all steps in one launch pair, and the whole backward -- eager and replayed from one HIP graph.  The
    recurrence (LSTM, feature nets, decoder) is the host project's and is not here: its outputs are given tensors,
    and the decoder's d loss / d z arrives as a given gradient."""
    from vae_amd import vrnn as V
    from vae_amd.reparameterizers import IsotropicGaussian
    cfg = {'reparam_type': 'isotropic_gaussian', 'continuous_size': 2 * L, 'nll_type': 'bernoulli', 'kl_beta': 1.0,
           'use_prior_kl': True, 'cuda': True, 'half': False, 'continuous_mut_info': 0, 'discrete_mut_info': 0}
    rep = IsotropicGaussian(cfg).train()
    xs = [(T(B, 1, 28, 28, kind='rand') < 0.3).float() for _ in range(Tn)]
    gz = T(B, L) * (1.0 / (Tn * B))

    def leaves():
        return ([T(B, 2 * L).requires_grad_(True) for _ in range(Tn)], [T(B, 2 * L).requires_grad_(True) for _ in range(Tn)],
                [T(B, 1, 28, 28).requires_grad_(True) for _ in range(Tn)])

    def make(enc, pri, recons):
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
        return fn
    # pair fwd 40L+4 and bwd 48L+8 per step and sample; loss fwd 8D+16, bwd 12D
    nbytes = Tn * B * ((40 * L + 4) + (48 * L + 8) + (8 * D + 16) + 12 * D)
    cfgs = 'T=%d B=%d L=%d %dpx bernoulli f32: %d pair steps + loss over all steps + backward' % (Tn, B, L, D, Tn)
    timeit('vrnn sequence (eager)', cfgs, make(*leaves()), nbytes)
    # fresh leaves whose first backward runs on the capture's side stream (see concat_module)
    fn = make(*leaves())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    timeit('vrnn sequence (HIP graph)', cfgs, graph.replay, nbytes)


def c4_module(B, L=128, F=256, C=3, H=64, W=64):
    """BASELINE config 4 through the drop-in classes: SequentialReparameterizer(['isotropic_gaussian', 'flow']) built
    from the reference's config keys (Gaussian L -> dense -> 5 x (MADE, BatchNormFlow, Reverse) of width F; the masked
    linears are stock GEMMs) and loss_function with the discretized-logistic NLL over 3x64x64 images, forward and
    backward, in-kernel noise.  Two lines: the latent chain alone (its kernels' bytes next to the GEMMs' time) and the
    loss alone (the step's byte mover: one pass, 20D + 12 bytes per sample)."""
    from vae_amd.reparameterizers import SequentialReparameterizer
    from vae_amd import elbo as E
    D = C * H * W
    cfg = {'continuous_size': 2 * L, 'discrete_size': 5, 'half': False, 'cuda': True, 'epochs': 10,
           'continuous_mut_info': 0, 'discrete_mut_info': 0, 'kl_beta': 1.0, 'use_prior_kl': False, 'flow_type': 'maf',
           'latent_size': F, 'memory_read_steps': 1, 'reparam_type': 'isotropic_gaussian', 'nll_type': 'disc_logistic'}
    torch.manual_seed(0)
    m = SequentialReparameterizer(['isotropic_gaussian', 'flow'], cfg).to(dev).train()
    logits = T(B, 2 * L).requires_grad_(True)
    x = (torch.floor(T(B, C, H, W, kind='rand') * 256) / 255.0).clamp(0, 1)
    recon = T(B, 2 * C, H, W).requires_grad_(True)
    Fw = m.output_size
    gzf, gkl = T(B, Fw) * (1.0 / B), torch.full((B,), 1.0 / B, device=dev)

    def chain():
        z, params = m(logits)
        torch.autograd.backward([z, m.kl(params)], [gzf, gkl])
        logits.grad = None
        for p in m.parameters():
            p.grad = None
    # our kernels on the chain: gauss fwd 12L+4 / bwd 20L+4; per MAF block made tail 16F+4 / 24F+4 and BatchNormFlow
    # 12F+8 / 20F+4; flow KL 4F+8 / 8F
    # (r03: the MADE tail leaves the BatchNormFlow moments: 8F+8 forward instead of 12F+8)
    chain_bytes = B * ((12 * L + 4) + (20 * L + 4) + 5 * ((16 * Fw + 4) + (24 * Fw + 4) + (8 * Fw + 8) + (20 * Fw + 4))
                       + (4 * Fw + 8) + 8 * Fw)
    what = ('B=%d Gaussian L=%d -> dense -> 5 x (MADE, BatchNormFlow, Reverse) F=%d, fwd+kl+bwd; bytes = the tails and '
            'KLs only, time includes the stock masked GEMMs' % (B, L, Fw))
    timeit('c4 latent chain (module)', what, chain, chain_bytes)
    # the same chain as ONE HIP graph (eager it is ~150 launches from Python).  Fresh leaves: a leaf that first
    # accumulated its gradient on the default stream would pull that stream into the capture.
    import copy
    m2 = copy.deepcopy(m)
    lg2 = logits.detach().clone().requires_grad_(True)
    leaves = [lg2] + [p for p in m2.parameters() if p.requires_grad]

    def chain2():
        z, params = m2(lg2)
        return torch.autograd.grad([z, m2.kl(params)], leaves, [gzf, gkl], allow_unused=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            chain2()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        keep = chain2()
    timeit('c4 latent chain (HIP graph)', what, graph.replay, chain_bytes)
    del graph, keep
    kld = T(B, kind='rand')

    def loss():
        out = E.loss_function(recon, x, kld, 'disc_logistic', kl_beta=1.0)
        out['loss_mean'].backward()
        recon.grad = None
    timeit('c4 loss (module)', 'B=%d %dx%dx%d disc_logistic f32: loss_function fwd (one pass: values + d recon) + backward'
           % (B, C, H, W), loss, B * (20 * D + 12 + 8))
def bernoulli(B, K):
    logits, U, gy = T(B, K), T(B, K, kind='rand'), T(B, K)
    y, zh, gl, kl, gkl = T(B, K, kind='e'), T(B, K, kind='e'), T(B, K, kind='e'), T(B, kind='e'), T(B)
    cfg = 'B=%d K=%d f32' % (B, K)
    timeit('bernoulli_fwd', cfg, lambda: call('vae_bernoulli_fwd', ptr(logits), K, ptr(U), ptr(y), 0, ptr(zh), 0, B, K, 0.7, 0, 0, 0, 0, st), B * 16 * K)
    timeit('bernoulli_bwd', cfg, lambda: call('vae_bernoulli_bwd', ptr(logits), K, ptr(y), 0, ptr(gy), 0, None, 0, ptr(gl), 0, B, K, 0.7, 0, st), B * 16 * K)
    timeit('bernoulli_kl_fwd', cfg, lambda: call('vae_bernoulli_kl_fwd', ptr(logits), K, None, 0, ptr(kl), B, K, 0, st), B * (4 * K + 4))
    timeit('bernoulli_kl_bwd', cfg, lambda: call('vae_bernoulli_kl_bwd', ptr(logits), K, None, 0, ptr(gkl), ptr(gl), None, B, K, 0, st), B * (8 * K + 4))


def beta(B, L):
    logits = T(B, 2 * L)
    c1, c0, kl, gkl, gl = T(B, L, kind='e'), T(B, L, kind='e'), T(B, kind='e'), T(B), T(B, 2 * L, kind='e')
    g1 = T(B, L)
    means = torch.zeros(2, device=dev)
    cfg = 'B=%d L=%d f32' % (B, L)
    timeit('beta_split_kl_fwd', cfg, lambda: call('vae_beta_split_kl_fwd', ptr(logits), 2 * L, ptr(c1), ptr(c0), ptr(kl), ptr(means), ptr(ws), B, L, 1.0 / (B * L), 0, st), B * (16 * L + 4))
    timeit('beta_split_kl_bwd', cfg, lambda: call('vae_beta_split_kl_bwd', ptr(logits), 2 * L, ptr(g1), ptr(g1), ptr(gkl), None, ptr(gl), B, L, 1.0 / (B * L), 0, st), B * (24 * L + 4))


def flow(B, F):
    u, ld, kl, gu = T(B, F), T(B, 1), T(B, kind='e'), T(B, F, kind='e')
    mean, gsum = torch.zeros(1, device=dev), torch.ones(1, device=dev)
    cfg = 'B=%d F=%d f32 cross-batch' % (B, F)
    timeit('flow_kl_fwd', cfg, lambda: call('vae_flow_kl_fwd', ptr(u), ptr(ld), 1, ptr(kl), ptr(mean), ptr(ws), B, F, 1.0 / B, 1, 0, 0, st), B * (4 * F + 8))
    timeit('flow_kl_bwd', cfg, lambda: call('vae_flow_kl_bwd', ptr(u), None, ptr(gsum), ptr(gu), B, F, 1.0 / B, 0, st), B * 8 * F)


def nll(B, D, code, name, dtype=torch.float32):
    es = 2 if dtype == torch.bfloat16 else 4
    dc = 1 if dtype == torch.bfloat16 else 0
    R = D if code == 0 else 2 * D
    x = (T(B, D, kind='rand') * 255).floor().div(255).to(dtype) if code else (T(B, D, kind='rand') < 0.13).to(dtype)
    recon, kld = T(B, R, dtype=dtype), T(B, dtype=dtype, kind='rand')
    n, lo, el, gr, gk = T(B, dtype=dtype, kind='e'), T(B, dtype=dtype, kind='e'), T(B, dtype=dtype, kind='e'), T(B, R, dtype=dtype, kind='e'), T(B, dtype=dtype, kind='e')
    means, gm = torch.zeros(6, device=dev), torch.zeros(6, device=dev)
    gm[0] = 1
    cfg = 'B=%d D=%d %s %s' % (B, D, name, 'bf16' if dc else 'f32')
    timeit('nll_elbo_fwd', cfg, lambda: call('vae_nll_elbo_fwd', ptr(x), ptr(recon), code, ptr(kld), None, None, 1.0, 1.0 / B, ptr(n), ptr(lo), ptr(el), ptr(means), ptr(ws), B, D, dc, st), B * ((D + R) * es + 3 * es))
    timeit('nll_elbo_bwd', cfg, lambda: call('vae_nll_elbo_bwd', ptr(x), ptr(recon), code, None, None, None, ptr(gm), 1.0, 1.0 / B, ptr(gr), ptr(gk), None, None, B, D, dc, st), B * (D + 2 * R) * es)


def nll_fused(B, D, code, name, dtype=torch.float32):
    R = D if code == 0 else 2 * D
    es = 2 if dtype == torch.bfloat16 else 4
    dc = 1 if dtype == torch.bfloat16 else 0
    x = (T(B, D, kind='rand') * 255).floor().div(255) if code else (T(B, D, kind='rand') < 0.13).float()
    x = x.to(dtype)
    recon, kld = T(B, R, dtype=dtype), T(B, kind='rand', dtype=dtype)
    n, lo, el, gr = T(B, kind='e', dtype=dtype), T(B, kind='e', dtype=dtype), T(B, kind='e', dtype=dtype), T(B, R, kind='e', dtype=dtype)
    means, gm = torch.zeros(6, device=dev), torch.zeros(6, device=dev)
    gm[0] = 1
    cfg = 'B=%d D=%d %s %s (fwd + gradient, one pass)' % (B, D, name, 'bf16' if dc else 'f32')
    timeit('nll_elbo_fused', cfg, lambda: call('vae_nll_elbo_fused', ptr(x), ptr(recon), code, ptr(kld), None, None, 1.0, 1.0 / B, 1.0, ptr(n), ptr(lo), ptr(el), ptr(means), ptr(ws), ptr(gr), B, D, dc, st), B * ((D + 2 * R) * es + 3 * es))
    timeit('nll_fused_grad_finish', cfg, lambda: call('vae_nll_fused_grad_finish', ptr(gr), ptr(gm), B * R, dc, st), 12)


def flowtail(B, F):
    """One MAF block's tails: MADE (x, [m|a] -> u, logdet) then BatchNormFlow + Reverse (train)."""
    x, ma, gu, gld = T(B, F), T(B, 2 * F) * 0.3, T(B, F), T(B)
    u, ld, gx, gma = T(B, F, kind='e'), T(B, kind='e'), T(B, F, kind='e'), T(B, 2 * F, kind='e')
    off = lambda t, n: ctypes.c_void_p(t.data_ptr() + 4 * n)
    cfg = 'B=%d F=%d f32' % (B, F)
    timeit('made_tail_fwd', cfg, lambda: call('vae_made_tail_fwd', ptr(x), F, ptr(ma), 2 * F, off(ma, F), 2 * F, ptr(u), None, ptr(ld), B, F, 0, 0, st), B * (16 * F + 4))
    timeit('made_tail_bwd', cfg, lambda: call('vae_made_tail_bwd', ptr(u), off(ma, F), 2 * F, ptr(gu), ptr(gld), ptr(gx), ptr(gma), 2 * F, off(gma, F), 2 * F, B, F, 0, 0, st), B * (24 * F + 4))
    lg, beta = T(F) * 0.1, T(F)
    stats = torch.empty(_lib.load().vae_bnflow_stats_floats(F), device=dev)
    y, ld2 = T(B, F, kind='e'), T(B, kind='e')
    timeit('bnflow_fwd (stats+apply+reverse)', cfg, lambda: call('vae_bnflow_fwd', ptr(u), ptr(lg), ptr(beta), None, None, ptr(stats), ptr(y), ptr(ld), ptr(ld2), B, F, 1e-5, 1, 1, 0, st), B * (12 * F + 8))
    if _lib.load().vae_made_tail_moments_supported(F, 0):
        mom = lambda: call('vae_made_tail_fwd_moments', ptr(x), F, ptr(ma), 2 * F, off(ma, F), 2 * F, ptr(u), None, ptr(ld), ptr(stats), B, F, 0, 0, st)
        bn2 = lambda: call('vae_bnflow_fwd', ptr(u), ptr(lg), ptr(beta), None, None, ptr(stats), ptr(y), ptr(ld), ptr(ld2), B, F, 1e-5, 2, 1, 0, st)
        timeit('made_tail_fwd_moments (u + column moments)', cfg, mom, B * (16 * F + 4))
        timeit('bnflow_fwd (moments given: apply+reverse)', cfg, bn2, B * (8 * F + 8))
        timeit('MAF block fwd tails, folded (2 launches)', cfg, lambda: (mom(), bn2()), B * (24 * F + 12))
        timeit('MAF block fwd tails, plain (2 launches)', cfg, lambda: (
            call('vae_made_tail_fwd', ptr(x), F, ptr(ma), 2 * F, off(ma, F), 2 * F, ptr(u), None, ptr(ld), B, F, 0, 0, st),
            call('vae_bnflow_fwd', ptr(u), ptr(lg), ptr(beta), None, None, ptr(stats), ptr(y), ptr(ld), ptr(ld2), B, F, 1e-5, 1, 1, 0, st)),
            B * (28 * F + 12))
    timeit('bnflow_bwd (stats+apply)', cfg, lambda: call('vae_bnflow_bwd', ptr(u), ptr(gu), ptr(gld), ptr(stats), ptr(gx), B, F, 1, 1, 0, st), B * (20 * F + 4))
    s_raw, t_raw, mask = T(B, F) * 0.3, T(B, F), (torch.arange(F, device=dev) % 2).float()
    gs, gt = T(B, F, kind='e'), T(B, F, kind='e')
    timeit('coupling_tail_fwd', cfg, lambda: call('vae_coupling_tail_fwd', ptr(x), ptr(s_raw), ptr(t_raw), ptr(mask), ptr(y), None, ptr(ld), B, F, 0, st), B * (16 * F + 4))
    timeit('coupling_tail_bwd', cfg, lambda: call('vae_coupling_tail_bwd', ptr(x), ptr(s_raw), ptr(mask), ptr(gu), ptr(gld), ptr(gx), ptr(gs), ptr(gt), B, F, 0, st), B * (24 * F + 4))


def graph_step(B, L):
    """The whole module-level ELBO step (IsotropicGaussian + Bernoulli loss_function + autograd backward): eager
    launches from Python vs one HIP graph replay (launch-bound at the small configs)."""
    from vae_amd.reparameterizers import IsotropicGaussian
    from vae_amd.elbo import loss_function
    m = IsotropicGaussian({'continuous_size': 2 * L, 'cuda': True, 'half': False})
    m.train()
    lg = T(B, 2 * L).requires_grad_(True)
    rc = T(B, 1, 28, 28).requires_grad_(True)
    xi = (T(B, 1, 28, 28, kind='rand') < 0.13).float()
    gz = T(B, L)

    def step():
        z, params = m(lg)
        out = loss_function(rc, xi, m.kl(params), 'bernoulli')
        return torch.autograd.grad([out['loss_mean'], z], [lg, rc], [None, gz], allow_unused=True)

    nbytes = B * ((12 * L + 4) + (20 * L + 4) + (12 * 784 + 12))
    cfg = 'B=%d L=%d 1x28x28 f32 module fwd+loss+bwd' % (B, L)
    timeit('elbo step (eager)', cfg, step, nbytes)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    timeit('elbo step (HIP graph)', cfg, graph.replay, nbytes)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--json', default=None)
    ap.add_argument('--loop', action='store_true', help='time every launch as a back-to-back loop (warm caches)')
    ap.add_argument('--sections', default='', help='comma list of: c2 c1 c3 c4 tails c5 multi bernbeta (default: all)')
    args = ap.parse_args()
    want = [k for k in args.sections.split(',') if k]
    if not want or 'c2' in want:
        print('== C2 family: IsotropicGaussian + Bernoulli NLL sweeps')
        for B, L in [(4096, 32), (16384, 128), (65536, 512), (65536, 1024)]:
            gauss(B, L)
        gauss(65536, 512, philox=True)
        gauss(65536, 512, dtype=torch.bfloat16)
        for B in (4096, 16384, 65536):
            nll(B, 784, 0, 'bernoulli')
        nll(65536, 784, 0, 'bernoulli', dtype=torch.bfloat16)
        nll_fused(65536, 784, 0, 'bernoulli')
        nll_fused(65536, 784, 0, 'bernoulli', dtype=torch.bfloat16)
    if not want or 'c1' in want:
        print('== C1-sized steps: eager launches vs one HIP graph')
        graph_step(4096, 32)
        graph_step(16384, 128)
        graph_step(65536, 512)
    if not want or 'c3' in want:
        print('== C3: Gumbel-softmax / Concat')
        for K in (32, 256, 1024):
            gumbel(65536, K)
        gumbel(65536, 256, philox=True)
        concat(65536, 128, 256)
        concat_module(65536, 128, 256)
    if not want or 'c4' in want:
        print('== C4: Gaussian->Flow, 3x64x64 discretized logistic')
        nll(8192, 12288, 2, 'disc_logistic')
        nll(8192, 12288, 1, 'gaussian')
        flow(65536, 512)
        nll_fused(8192, 12288, 2, 'disc_logistic')
        nll_fused(8192, 12288, 1, 'gaussian')
        c4_module(8192)
    if not want or 'tails' in want:
        print('== flow-layer tails (one MAF / RealNVP block)')
        flowtail(65536, 512)
        flowtail(65536, 64)
    if not want or 'c5' in want:
        print('== C5: VRNN per-step pair (B per step) and the time-batched loss (T*B rows)')
        gauss(16384, 128)
        gauss_pair(16384, 128)
        pair_fused(16384, 128)
        pair_fused(4096, 32)
        nll(64 * 1024, 784, 0, 'bernoulli T=64 x B=1024')
        nll_steps(64, 1024, 784)
        vrnn_sequence(64, 1024, 128)
        vrnn_sequence(64, 4096, 128)
    if not want or 'multi' in want:
        print('== multi-sample forms (loss_function K > 1, MSGVAE, likelihood)')
        multi(65536, 512, 784, 4)
        multi(16384, 128, 784, 8)
        likelihood(64, 1024)
    if not want or 'bernbeta' in want:
        print('== Bernoulli / Beta reparameterizers')
        bernoulli(65536, 1024)
        beta(65536, 512)
    if args.json:
        json.dump(rows, open(args.json, 'w'), indent=1)
