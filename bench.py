#!/usr/bin/env python3
"""ELBO-step throughput of the hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): simple_vae, IsotropicGaussian latent 512, B = 65536 samples
per GPU, 1x28x28 Bernoulli NLL, fp32, synthetic inputs resident in HBM.  One step = one pass of
the hot path over one batch, forward and backward:

    gauss_reparam_kl_fwd -> nll_elbo_fused (forward AND d loss_mean / d recon in one read of x, recon; the six
    batch means by a <= 64-block closing launch inside the same C-ABI call, bracketed with it)
    -> [N>1: one packed all-reduce of the 8 batch scalars] -> nll_fused_backward (one launch: device-side
    rescale, a no-op for loss_mean.backward()'s unit gradient, and d kld [B]) -> gauss_reparam_kl_bwd

(`--two-pass` times the reference-shaped sequence instead: nll_elbo_fwd ... nll_elbo_bwd, which reads x and
recon twice; both produce the same outputs and gradients.)

The encoder/decoder networks are not part of the path (stock PyTorch in the reference's host
project); their outputs (encoder logits, decoder logits) and the decoder's gradient w.r.t. z are
synthetic tensors.  Every kernel goes through the C ABI (libvae_amd.so); HIP events around every
launch of the timed steps give the per-kernel durations behind `roofline`.

The batch shards by rows with no data-path collective except the scalar all-reduce.  `value` is the WEAK-scaling
figure (B = 65536 per GPU); `strong_scaling` in the same line is the other reading of "the batch is sharded across the
8 GPUs": a GLOBAL batch of 65536, B / N rows per GPU, the four launches replayed from one HIP graph (at 8192 rows per
GPU the step is launch-bound otherwise) with the scalar all-reduce beside it.  `configs` holds BASELINE.json's other
configurations (C3 Concat, C4 Gaussian->Flow + discretized logistic, C5 VRNN T = 64) through the module classes, timed
by scripts/bench_configs.py's functions.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from vae_amd.timing import Event as _Event  # noqa: E402  (HIP events without a system-scope fence)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
B_PER_GPU, L, C, H, W = 65536, 512, 1, 28, 28
D = C * H * W


def algorithmic_bytes(B, philox=False):
    """SURVEY 8(d), fp32, per launch (the noise terms drop out with in-kernel Philox)."""
    nz = 0 if philox else 4 * L
    return {
        'gauss_reparam_kl_fwd': (12 * L + nz + 4) * B,   # logits 8L + noise 4L + z 4L + kl 4
        'gauss_reparam_kl_bwd': (20 * L + nz + 4) * B,   # logits 8L + noise 4L + gz 4L + gkl 4 + glogits 8L
        'nll_elbo_fwd': (8 * D + 12) * B,           # x 4D + recon 4D + kld 4 + loss, elbo 8
        'nll_elbo_bwd': (12 * D) * B,               # x 4D + recon 4D + grecon 4D
        'nll_elbo_fused': (12 * D + 12) * B,        # x 4D + recon 4D + grecon 4D + kld 4 + loss, elbo 8
        'nll_fused_backward': 4 * B,                # gmeans -> d kld [B] (+ the skipped rescale): latency only
    }


def cpu_baseline(seconds=10.0, L=L, label='configs[1] shapes'):
    """The oracle's unfused op sequence (the reference's CPU path restated) on the host cores."""
    from oracle import cpu_ref as O
    threads = torch.get_num_threads()
    Bc = 4096
    g = torch.Generator().manual_seed(1234)
    logits = torch.randn(Bc, 2 * L, generator=g)
    noise = torch.randn(Bc, L, generator=g)
    x = (torch.rand(Bc, C, H, W, generator=g) < 0.13).float()
    recon = torch.randn(Bc, C, H, W, generator=g)
    O.elbo_step(logits, noise, x, recon)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        O.elbo_step(logits, noise, x, recon)
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds:
            break
    return {'value': Bc * n / dt, 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
            'sample': '%s: %d fwd+bwd ELBO steps of B=%d, L=%d, 1x28x28 Bernoulli, fp32, oracle/cpu_ref.elbo_step '
                      '(torch CPU ops in the reference op order), %.1f s' % (label, n, Bc, L, dt)}


def configs_block():
    """BASELINE.json's other configurations through the drop-in classes, timed by the functions of
    scripts/bench_configs.py (each launch sequence behind a 512 MiB fill of another buffer, median of 9: cache-cold,
    what a training step sees; profiles/r03_configs.txt is the same functions run from that script)."""
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import bench_configs as bc
    bc.QUIET = True
    bc.rows.clear()
    bc.concat_module(65536, 128, 256)          # C3
    bc.c4_module(8192)                         # C4
    bc.vrnn_sequence(64, 4096, 128)            # C5
    keep = ('concat step (eager)', 'concat step (HIP graph)', 'c4 latent chain (module)', 'c4 latent chain (HIP graph)',
            'c4 loss (module)',
            'vrnn sequence (eager)', 'vrnn sequence (HIP graph)')
    tag = {'concat': 'configs[2] C3', 'c4': 'configs[3] C4', 'vrnn': 'configs[4] C5'}
    out = []
    for r in bc.rows:
        if r['kernel'] in keep:
            out.append({'config': tag[r['kernel'].split()[0]], 'what': r['kernel'], 'shape': r['config'],
                        'ms_per_step': r['ms'], 'algorithmic_bytes': int(r['algorithmic_MB'] * 1e6),
                        'achieved_GBs': r['GBs'], 'frac': r['frac_of_8TBs']})
    bc.rows.clear()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--repeats', type=int, default=5,
                    help='timed regions of --steps steps each (every one fenced on both sides); value = the median one')
    ap.add_argument('--batch', type=int, default=B_PER_GPU, help='samples per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help="skip the `configs` block (BASELINE.json's C3 / C4 / C5)")
    ap.add_argument('--main-only', action='store_true',
                    help='only the timed region (profiling runs: no Philox variant, no module path, no CPU leg)')
    ap.add_argument('--two-pass', action='store_true', help='separate NLL forward and backward kernels')
    ap.add_argument('--materialize-logvar', action='store_true',
                    help="also write params['gaussian']['logvar'] in the forward (the module classes produce it on first "
                         "read instead: nothing in the training step reads it)")
    ap.add_argument('--noise', choices=['injected', 'philox'], default='injected',
                    help='injected: eps read from HBM (the parity-test form); philox: drawn inside the kernels')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                             % (args.gpus, args.gpus))
    # one rank per GPU; BENCH_BACKEND=gloo + fewer GPUs than ranks is a rehearsal mode for 1-GPU boxes only
    backend = os.environ.get('BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if backend == 'nccl' and local_rank >= ndev:
        raise SystemExit('bench.py: rank %d has no GPU (%d visible)' % (local_rank, ndev))
    dev = torch.device('cuda', local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    dist = None
    # BENCH_FORCE_DIST=1: take the N > 1 code path (process group, async all-reduces, strong leg) with a single rank --
    # the only way to run the RCCL side of this file on a one-GPU box (torch.distributed.run --nproc-per-node 1)
    if world > 1 or os.environ.get('BENCH_FORCE_DIST', '0') == '1':
        import torch.distributed as dist
        if backend == 'nccl':
            # the collective's stream at high priority: its one small kernel is then scheduled beside a backward launch
            # that fills every CU instead of behind it (one rank under RCCL, scripts/rccl_one_rank.sh: step 0.348 ->
            # 0.340 ms with the all-reduce overlapped mid-step; 0.325 without any collective, 0.337 synchronous)
            os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '1')
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    def all_reduce(t, op=None):
        # RCCL reduces device tensors in place; the gloo rehearsal goes through the host
        op = dist.ReduceOp.SUM if op is None else op
        if backend == 'nccl':
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    from vae_amd import _lib
    from vae_amd._lib import call, ptr
    _lib.load()

    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    f32 = dict(dtype=torch.float32, device=dev)
    logits = torch.randn(B, 2 * L, generator=gen, **f32)
    noise = torch.randn(B, L, generator=gen, **f32)
    x = (torch.rand(B, D, generator=gen, **f32) < 0.13).float()
    recon = torch.randn(B, D, generator=gen, **f32)
    gz = torch.randn(B, L, generator=gen, **f32) * (1.0 / B)   # the decoder's d loss / d z (synthetic)
    # outputs, allocated once
    z = torch.empty(B, L, **f32)
    logvar = torch.empty(B, L, **f32)
    kl = torch.empty(B, **f32)
    nll = torch.empty(B, **f32)
    loss = torch.empty(B, **f32)
    elbo = torch.empty(B, **f32)
    scalars = torch.zeros(8, **f32)          # 6 ELBO means + mu_mean, logvar_mean: one packed message
    gmeans = torch.zeros(6, **f32)
    gmeans[0] = 1.0                           # d(loss_mean) = 1: what loss_mean.backward() seeds
    grecon = torch.empty(B, D, **f32)
    gkld = torch.empty(B, **f32)
    glogits = torch.empty(B, 2 * L, **f32)
    ws = _lib.workspace(dev)
    st = _lib.stream_ptr(dev)
    wsst = [ws, st]          # rebound to the capture stream's pair for the graph-replay variant
    inv_count = 1.0 / (B * world)
    mean_scale = 1.0 / (B * world * L)
    row0 = rank * B

    fused = not args.two_pass
    philox = args.noise == 'philox'
    plogvar = ptr(logvar) if args.materialize_logvar else None
    noise_arg = [None if philox else ptr(noise)]   # None: the kernels draw Philox noise themselves
    names = ['gauss_reparam_kl_fwd', 'nll_elbo_fused', 'nll_fused_backward', 'gauss_reparam_kl_bwd'] if fused else \
        ['gauss_reparam_kl_fwd', 'nll_elbo_fwd', 'nll_elbo_bwd', 'gauss_reparam_kl_bwd']
    pmeans = ctypes_off(scalars, 6)
    T = {'logits': logits, 'noise': noise, 'x': x, 'recon': recon, 'gz': gz, 'z': z, 'kl': kl, 'nll': nll, 'loss': loss,
         'elbo': elbo, 'scalars': scalars, 'gmeans': gmeans, 'grecon': grecon, 'gkld': gkld, 'glogits': glogits,
         'pmeans': pmeans, 'dt': 0}        # what step() runs on: swapped for the 16-bit variant below

    solo = world == 1 and dist is None        # the single-process extras (variants, graph captures, configs, CPU leg)
    overlap = dist is not None and backend == 'nccl' and os.environ.get('BENCH_SYNC_ALLREDUCE', '0') != '1'
    # BENCH_ALLREDUCE=deferred: the collective is enqueued at the END of the step on a copy of the scalars and joined
    # at the end of the NEXT step (no cross-stream event in the middle of a step; a whole step to complete in)
    deferred = overlap and os.environ.get('BENCH_ALLREDUCE', 'midstep') == 'deferred'
    scal_copy = torch.zeros_like(scalars)
    pending = [None]
    step_no = [1]   # Philox call offset: a fresh draw every step, the same one in that step's backward
    rng_cell = [None]   # graph replay with in-kernel draws: the offset lives in this device word (vae_rng_next)

    def step(ev=None):
        def mark(i):
            if ev is not None:
                ev[i].record()
        ws, st = wsst
        off = step_no[0]
        if rng_cell[0] is not None:
            call('vae_rng_next', ptr(rng_cell[0]), 1, st)
            off = (1 << 63) | rng_cell[0].data_ptr()          # VAE_OFFSET_ON_DEVICE
        mark(0)
        call('vae_gauss_reparam_kl_fwd', ptr(T['logits']), 2 * L, noise_arg[0], ptr(T['z']), L, plogvar, ptr(T['kl']), T['pmeans'],
             ptr(ws), B, L, 1e-6, mean_scale, 1, 1234, off, row0, T['dt'], st)
        mark(1)
        if fused:
            call('vae_nll_elbo_fused', ptr(T['x']), ptr(T['recon']), 0, ptr(T['kl']), None, None, 1.0, inv_count, 1.0, ptr(T['nll']),
                 ptr(T['loss']), ptr(T['elbo']), ptr(T['scalars']), ptr(ws), ptr(T['grecon']), B, D, T['dt'], st)
        else:
            call('vae_nll_elbo_fwd', ptr(T['x']), ptr(T['recon']), 0, ptr(T['kl']), None, None, 1.0, inv_count, ptr(T['nll']), ptr(T['loss']),
                 ptr(T['elbo']), ptr(T['scalars']), ptr(ws), B, D, T['dt'], st)
        mark(2)
        if dist is not None:
            # global-batch means (abstract_vae.py:381-387) in one 8-float message.  Nothing later in the step reads
            # them (the kernels pre-scale by 1/B_global), so over RCCL the collective runs on its own stream beside
            # the backward kernels and is joined at the end of the step.
            if deferred:
                pass
            elif overlap:
                pending[0] = dist.all_reduce(T['scalars'], op=dist.ReduceOp.SUM, async_op=True)
            else:
                all_reduce(T['scalars'])
        mark(3)
        if fused:
            call('vae_nll_fused_backward', ptr(T['grecon']), B * D, ptr(T['gmeans']), 1.0, inv_count, ptr(T['gkld']), None, None, B,
                 T['dt'], st)
        else:
            call('vae_nll_elbo_bwd', ptr(T['x']), ptr(T['recon']), 0, None, None, None, ptr(T['gmeans']), 1.0, inv_count,
                 ptr(T['grecon']), ptr(T['gkld']), None, None, B, D, T['dt'], st)
        mark(4)
        call('vae_gauss_reparam_kl_bwd', ptr(T['logits']), 2 * L, noise_arg[0], ptr(T['gz']), L, None, ptr(T['gkld']), None,
             ptr(T['glogits']), 2 * L, B, L, 1e-6, mean_scale, 1, 1234, off, row0, T['dt'], st)
        mark(5)
        if pending[0] is not None:
            pending[0].wait()                 # the launch stream waits for the collective; no host block
            pending[0] = None
        if deferred:
            scal_copy.copy_(T['scalars'].float())   # behind the wait: the previous collective is done with scal_copy
            pending[0] = dist.all_reduce(scal_copy, op=dist.ReduceOp.SUM, async_op=True)
        step_no[0] += 1

    for _ in range(args.warmup):
        step()
    # Per-kernel durations: HIP events around every launch of every step of ONE extra region of K steps, run before the
    # timed regions and kept out of `value` (an event record is a barrier packet: bracketing all four launches costs
    # ~15 us of a 326 us step).  BENCH_EVENT_EVERY=n brackets every n-th step of that region only.
    every = max(1, int(os.environ.get('BENCH_EVENT_EVERY', '1')))
    events = [[_Event() for _ in range(6)] if k % every == 0 else None
              for k in range(args.steps)]

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # the bracketed region (kernel durations only)
    fence()
    for k in range(args.steps):
        step(events[k])
    # R timed regions of exactly K steps, each fenced (barrier + synchronize) on both sides, MAX over ranks per region;
    # the figure is the MEDIAN region (min / max beside it): one 7-18 ms sample is thin evidence.
    region_ms = []
    for r in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step()
        fence()
        dtr = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dtr], dtype=torch.float64, device=dev)
            all_reduce(t, op=dist.ReduceOp.MAX)
            dtr = float(t.item())
        region_ms.append(dtr * 1e3)
    dt = sorted(region_ms)[len(region_ms) // 2] * 1e-3

    # per-kernel average durations from the HIP events recorded on the launch stream during the timed steps
    seg = {'gauss_reparam_kl_fwd': (0, 1), names[1]: (1, 2), names[2]: (3, 4), 'gauss_reparam_kl_bwd': (4, 5)}
    timed = [e for e in events if e is not None]
    # median over the bracketed steps: an event record is a barrier packet, and the odd bracket that lands behind a
    # host hiccup is not the kernel's duration (rocprofv3's kernel-trace averages for the same command are committed
    # under profiles/ for comparison)
    import statistics
    dur_ms = {n: statistics.median(e[a].elapsed_time(e[b]) for e in timed) for n, (a, b) in seg.items()}
    if pending[0] is not None:
        pending[0].wait()
        pending[0] = None
    sanity = float((scal_copy if deferred else scalars)[0].item())
    if not (sanity == sanity):
        raise SystemExit('bench: loss_mean is NaN')

    # SURVEY 8(d)'s other figure, the one "north_star"'s 1 -> 8 GPU scaling target is about: the WHOLE model step --
    # stock encoder / decoder (the example's nn.Linear 784-512-2L / L-512-784 networks, Adam) around the drop-in
    # latent + loss half, at the same C2 shapes, with everything that crosses ranks inside the timed region: the packed
    # all-reduce of the loss scalars (vae_amd.dist.globalize_loss) and ONE flat all-reduce of the network gradients
    # (vae_amd.dist.FlatGradients; RCCL over xGMI under the default backend).  Reported beside the hot-path figure.
    whole_model = None
    strong = None
    if not args.main_only:
        whole_model = whole_model_leg(dev, x.view(B, C, H, W), B, world, fence, all_reduce if dist is not None else None)
        whole_model['scaling'] = 'weak'
        # the strong-scaling leg: global batch 65536 over the ranks (at one GPU it is the same batch as the weak leg,
        # replayed from a graph)
        if B == B_PER_GPU:
            strong = strong_leg(dev, rank, world, dist, backend, fence, all_reduce, args.steps, args.warmup, args.repeats,
                                use_graph=os.environ.get('BENCH_STRONG_GRAPH', '1') != '0')
            if world > 1:
                _, Bs = __import__('vae_amd.dist', fromlist=['shard_rows']).shard_rows(B_PER_GPU, rank, world)
                wm = whole_model_leg(dev, x.view(B, C, H, W)[:Bs], Bs, world, fence, all_reduce, global_batch=B_PER_GPU)
                wm['scaling'] = 'strong'
                strong['whole_model'] = wm
            else:
                strong['whole_model'] = 'one GPU: identical to whole_model'

    # the production form of the same step (the module classes' default): eps drawn inside the kernels from
    # Philox4x32-10 instead of read from HBM -- reported beside the main figure, never as `value`
    philox_variant = None
    if solo and not philox and not args.main_only:
        noise_arg[0] = None
        for _ in range(args.warmup):
            step()
        ev2 = [[_Event() for _ in range(6)] if k % every == 0 else None
               for k in range(args.steps)]
        for k in range(args.steps):
            step(ev2[k])
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for k in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        dt2 = time.perf_counter() - t1
        noise_arg[0] = ptr(noise)
        ab2 = algorithmic_bytes(B, True)
        philox_variant = {'value': B * args.steps / dt2, 'unit': 'samples/s', 'ms_per_step': dt2 / args.steps * 1e3,
                          'kernels': {}}
        for n, (a, b) in seg.items():
            t2e = [e for e in ev2 if e is not None]
            ms = statistics.median(e[a].elapsed_time(e[b]) for e in t2e)
            philox_variant['kernels'][n] = {'avg_ms': round(ms, 5), 'algorithmic_bytes': ab2[n],
                                            'frac': round(ab2[n] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # the same step on bf16 tensors (north_star's second element type; arithmetic stays fp32 inside the kernels):
    # reported beside the figure, never as `value` (BASELINE.json's metric is the fp32 step)
    bf16_variant = None
    if solo and not philox and not args.main_only and not args.materialize_logvar:
        keep = dict(T)
        for k in ('logits', 'noise', 'x', 'recon', 'gz'):
            T[k] = keep[k].to(torch.bfloat16)
        for k in ('z', 'kl', 'nll', 'loss', 'elbo', 'grecon', 'gkld', 'glogits'):
            T[k] = torch.empty_like(keep[k], dtype=torch.bfloat16)
        T['dt'] = 1                                  # VAE_BF16
        noise_arg[0] = ptr(T['noise'])
        for _ in range(args.warmup):
            step()
        ev3 = [[_Event() for _ in range(6)] for _ in range(args.steps)]
        for k in range(args.steps):
            step(ev3[k])
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        for k in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        dt4 = time.perf_counter() - t3
        bf16_variant = {'value': B * args.steps / dt4, 'unit': 'samples/s', 'ms_per_step': dt4 / args.steps * 1e3,
                        'loss_mean': float(T['scalars'][0].item()), 'kernels': {}}
        for n, (a, b) in seg.items():
            ms = statistics.median(e[a].elapsed_time(e[b]) for e in ev3)
            nbytes = algorithmic_bytes(B)[n] // 2
            bf16_variant['kernels'][n] = {'avg_ms': round(ms, 5), 'algorithmic_bytes': nbytes,
                                          'frac': round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        T.clear()
        T.update(keep)
        noise_arg[0] = ptr(noise)
        del keep

    # the same launches recorded once into a HIP graph and replayed: what the launch gaps between the four kernels
    # cost.  Once with the injected noise, once with in-kernel Philox draws whose call offset lives in a device word
    # that a one-thread node of the graph advances (vae_rng_next), so every replay draws fresh noise.
    graph_replay = None
    if solo and not philox and not args.main_only:
        graph_replay = {}
        side = torch.cuda.Stream(device=dev)
        cell = torch.zeros(1, dtype=torch.int64, device=dev)
        for label in ('injected', 'philox'):
            noise_arg[0] = ptr(noise) if label == 'injected' else None
            rng_cell[0] = None if label == 'injected' else cell
            with torch.cuda.stream(side):
                wsst[0], wsst[1] = _lib.workspace(dev), _lib.stream_ptr(dev)
                step()                                   # warm the side stream's workspace outside the capture
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                wsst[1] = _lib.stream_ptr(dev)
                step()
            for _ in range(args.warmup):
                graph.replay()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            for _ in range(args.steps):
                graph.replay()
            torch.cuda.synchronize(dev)
            dt3 = time.perf_counter() - t2
            graph_replay[label] = {'value': B * args.steps / dt3, 'unit': 'samples/s',
                                   'ms_per_step': dt3 / args.steps * 1e3}
            del graph
        wsst[0], wsst[1] = ws, st
        noise_arg[0], rng_cell[0] = ptr(noise), None

    # the same step through the drop-in classes (autograd Functions, torch allocator): not the timed region,
    # reported beside it so the cost of the Python boundary is visible
    module_path = None
    if solo and not args.main_only:
        from vae_amd.reparameterizers import IsotropicGaussian
        from vae_amd.elbo import loss_function
        mod = IsotropicGaussian({'continuous_size': 2 * L, 'cuda': True, 'half': False})
        mod.train()
        lg = logits.detach().requires_grad_(True)
        rc = recon.view(B, C, H, W).detach().requires_grad_(True)
        xi = x.view(B, C, H, W)

        def mstep():
            zz, params = mod(lg, noise=noise)
            out = loss_function(rc, xi, mod.kl(params), 'bernoulli')
            torch.autograd.backward([out['loss_mean'], zz], [None, gz])
            lg.grad = None
            rc.grad = None
        for _ in range(5):
            mstep()
        e0, e1 = _Event(), _Event()
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(20):
            mstep()
        e1.record()
        torch.cuda.synchronize(dev)
        mms = e0.elapsed_time(e1) / 20
        module_path = {'ms_per_step': round(mms, 4), 'samples_per_s': B / (mms * 1e-3),
                       'what': 'IsotropicGaussian.forward + .kl + elbo.loss_function + autograd backward, 20 steps'}

    # the reference's own op sequence for this step in stock PyTorch-ROCm on this GPU (isotropic_gaussian.py:77-131,
    # 158-170 through torch.distributions; abstract_vae.py:356-387; Bernoulli-with-logits NLL), same resident inputs
    # and injected noise: what the reference does on an MI355X without this library.  Reported beside the figure.
    stock_torch = None
    if solo and not args.main_only:
        stock_torch = stock_torch_step_rate(logits, noise, x.view(B, C, H, W), recon.view(B, C, H, W), gz)

    if rank == 0:
        ab = algorithmic_bytes(B, philox)
        if args.materialize_logvar:
            ab['gauss_reparam_kl_fwd'] += 4 * L * B        # the logvar tensor is written too
        kernels = {}
        for n in names:
            gbs = ab[n] / (dur_ms[n] * 1e-3) / 1e9
            kernels[n] = {'avg_ms': round(dur_ms[n], 5), 'algorithmic_bytes': ab[n], 'achieved_GBs': round(gbs, 1),
                          'frac': round(gbs / HBM_PEAK_GBS, 4)}
        dom = max(names, key=lambda n: dur_ms[n])
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath) and B == B_PER_GPU:   # the PMC passes were collected at the default batch
            # profiles/traffic.json: PMC bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, scripts/summarize_pmc.py),
            # keyed by bench kernel name or by the full kernel symbol with a short 'kernel' field
            short = {'gauss_reparam_kl_fwd': 'gauss_fwd_kernel', 'gauss_reparam_kl_bwd': 'gauss_bwd_kernel',
                     'nll_elbo_fwd': 'nll_elbo_fwd_kernel', 'nll_elbo_bwd': 'nll_elbo_bwd_kernel',
                     'nll_elbo_fused': 'nll_elbo_fused_kernel'}.get(dom)
            try:
                tj = json.load(open(tpath))
                if isinstance(tj.get(dom), (int, float)):
                    traffic = tj[dom]
                else:
                    for v in tj.values():
                        if isinstance(v, dict) and v.get('kernel') == short:
                            traffic = int(v['hbm_bytes_per_launch'])
            except Exception:
                traffic = None
        total_bytes = sum(ab[n] for n in names)
        ksum = sum(dur_ms.values())
        out = {
            'metric': 'ELBO-step samples/sec', 'value': B * world * args.steps / dt, 'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'regions': {'n': len(region_ms), 'steps_each': args.steps, 'value_from': 'median',
                        'ms_per_step_min': min(region_ms) / args.steps, 'ms_per_step_max': max(region_ms) / args.steps},
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: simple_vae IsotropicGaussian latent 512, B=%d per GPU, 1x28x28 '
                                   'Bernoulli NLL; step = reparam+KL fwd, NLL+ELBO fwd, both backward (%s)'
                                   % (B, 'NLL forward and gradient in one pass' if fused else 'two-pass NLL'),
                       'global_batch': B * world, 'latent': L, 'pixels': D, 'parallelism': 'dp%d' % world,
                       'noise': 'in-kernel Philox4x32-10' if philox else 'injected (resident in HBM)',
                       'logvar_output': 'materialized' if args.materialize_logvar else 'on first read (not written)',
                       'kernel_timing': 'HIP events around each launch of %d steps of an extra, bracketed region of %d steps '
                                        '(not part of value), median' % (len(timed), args.steps)},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': kernels[dom]['achieved_GBs'],
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': kernels[dom]['frac'], 'traffic': traffic},
            'kernels': kernels,
            'hot_path': {'algorithmic_bytes_per_step': total_bytes, 'kernel_ms_per_step': round(ksum, 5),
                         'achieved_GBs': round(total_bytes / (ksum * 1e-3) / 1e9, 1),
                         'frac': round(total_bytes / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            'loss_mean': sanity,
            'whole_model': whole_model,
            'strong_scaling': strong,
            'module_path': module_path,
            'philox_variant': philox_variant,
            'bf16_variant': bf16_variant,
            'stock_torch_same_gpu': stock_torch,
            'graph_replay': graph_replay,
        }
        if solo and not args.main_only and not args.no_configs:
            out['configs'] = configs_block()
        if solo and not args.no_cpu_baseline and not args.main_only:
            out['cpu_baseline'] = cpu_baseline()
            # BASELINE.json configs[0], the designated CPU configuration: latent 32
            out['cpu_baseline_c1'] = cpu_baseline(seconds=6.0, L=32, label='configs[0] shapes')
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def strong_leg(dev, rank, world, dist, backend, fence, all_reduce, steps, warmup, repeats, use_graph=True):
    """The strong-scaling reading of north_star ("the batch is sharded across the 8 GPUs"): a GLOBAL batch of
    B_PER_GPU rows, this rank's contiguous shard of it (vae_amd.dist.shard_rows), the same four launches.  At 8192
    rows per GPU the launches are 5-20 us each and the step is launch-bound from Python, so the step is recorded once
    into a HIP graph and replayed; the 8-float all-reduce stays outside the graph, on RCCL's stream, on a COPY of the
    scalars (the next replay rewrites the originals while the collective may still be reading).
    Stream order per step: replay (launch stream) -> wait for the previous step's collective (launch stream waits, no
    host block) -> copy of the 8 scalars (launch stream) -> all-reduce enqueued: ProcessGroupNCCL makes its own stream
    wait for the launch stream's work so far, i.e. for that copy."""
    from vae_amd import _lib
    from vae_amd import dist as vdist
    from vae_amd._lib import call, ptr
    row0, B = vdist.shard_rows(B_PER_GPU, rank, world)
    gen = torch.Generator(device=dev).manual_seed(4321 + rank)
    f32 = dict(dtype=torch.float32, device=dev)
    logits = torch.randn(B, 2 * L, generator=gen, **f32)
    noise = torch.randn(B, L, generator=gen, **f32)
    x = (torch.rand(B, D, generator=gen, **f32) < 0.13).float()
    recon = torch.randn(B, D, generator=gen, **f32)
    gz = torch.randn(B, L, generator=gen, **f32) * (1.0 / B_PER_GPU)
    z, kl, nll, loss, elbo = (torch.empty(B, L, **f32), torch.empty(B, **f32), torch.empty(B, **f32),
                              torch.empty(B, **f32), torch.empty(B, **f32))
    scalars, scal_copy, gmeans = torch.zeros(8, **f32), torch.zeros(8, **f32), torch.zeros(6, **f32)
    gmeans[0] = 1.0
    grecon, gkld, glogits = torch.empty(B, D, **f32), torch.empty(B, **f32), torch.empty(B, 2 * L, **f32)
    inv_count, mean_scale = 1.0 / B_PER_GPU, 1.0 / (B_PER_GPU * L)
    pmeans = ctypes_off(scalars, 6)

    def launches(ws, st):
        call('vae_gauss_reparam_kl_fwd', ptr(logits), 2 * L, ptr(noise), ptr(z), L, None, ptr(kl), pmeans, ptr(ws), B, L,
             1e-6, mean_scale, 1, 1234, 1, row0, 0, st)
        call('vae_nll_elbo_fused', ptr(x), ptr(recon), 0, ptr(kl), None, None, 1.0, inv_count, 1.0, ptr(nll), ptr(loss),
             ptr(elbo), ptr(scalars), ptr(ws), ptr(grecon), B, D, 0, st)
        call('vae_nll_fused_backward', ptr(grecon), B * D, ptr(gmeans), 1.0, inv_count, ptr(gkld), None, None, B, 0, st)
        call('vae_gauss_reparam_kl_bwd', ptr(logits), 2 * L, ptr(noise), ptr(gz), L, None, ptr(gkld), None, ptr(glogits),
             2 * L, B, L, 1e-6, mean_scale, 1, 1234, 1, row0, 0, st)

    graph = None
    if use_graph:
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            launches(_lib.workspace(dev), _lib.stream_ptr(dev))      # the side stream's workspace, outside the capture
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            launches(_lib.workspace(dev), _lib.stream_ptr(dev))
    overlap = dist is not None and backend == 'nccl'
    pending = [None]

    def step():
        if graph is not None:
            graph.replay()
        else:
            launches(_lib.workspace(dev), _lib.stream_ptr(dev))
        if dist is not None:
            if pending[0] is not None:
                pending[0].wait()
                pending[0] = None
            scal_copy.copy_(scalars)
            if overlap:
                pending[0] = dist.all_reduce(scal_copy, op=dist.ReduceOp.SUM, async_op=True)
            else:
                all_reduce(scal_copy)

    for _ in range(warmup):
        step()
    region_ms = []
    for _ in range(max(1, repeats)):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None
        fence()
        dtr = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dtr], dtype=torch.float64, device=dev)
            all_reduce(t, op=dist.ReduceOp.MAX)
            dtr = float(t.item())
        region_ms.append(dtr * 1e3)
    dt = sorted(region_ms)[len(region_ms) // 2] * 1e-3
    loss_mean = float((scal_copy if dist is not None else scalars)[0].item())
    return {'scaling': 'strong', 'global_batch': B_PER_GPU, 'rows_this_rank': B, 'value': B_PER_GPU * steps / dt,
            'unit': 'samples/s', 'ms_per_step': dt / steps * 1e3, 'ms_per_step_min': min(region_ms) / steps,
            'ms_per_step_max': max(region_ms) / steps, 'launches': 'one HIP graph replay per step' if graph is not None
            else 'eager', 'loss_mean': loss_mean}


def whole_model_leg(dev, x_img, B, world, fence, all_reduce_max, steps=10, warmup=3, global_batch=None):
    import torch.distributed as tdist
    global_batch = B * world if global_batch is None else global_batch
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    from train_simple_vae import SimpleVAE
    from vae_amd import dist as vdist
    config = {'reparam_type': 'isotropic_gaussian', 'continuous_size': 2 * L, 'nll_type': 'bernoulli', 'kl_beta': 1.0,
              'cuda': True, 'half': False, 'continuous_mut_info': 0, 'discrete_mut_info': 0}
    torch.manual_seed(0)                       # the same initial weights on every rank
    model = SimpleVAE(config).to(dev).train()
    many = world > 1 or all_reduce_max is not None     # (a forced one-rank group: BENCH_FORCE_DIST)
    if many:
        vdist.configure(model, global_batch)
    # BENCH_GRAD_BUCKETS=n: n pieces, each sent as soon as backward has filled it (default: one message after backward;
    # 6.4 MB is latency-bound on xGMI either way)
    flat = vdist.FlatGradients(model.parameters(), buckets=int(os.environ.get('BENCH_GRAD_BUCKETS', '1')))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def wstep():
        recon, params = model(x_img)
        loss = model.loss_function(recon, x_img, params)
        if many:
            loss, _ = vdist.globalize_loss(loss)
        flat.zero()
        loss['loss_mean'].backward()
        flat.all_reduce()
        opt.step()
        return loss['loss_mean']
    for _ in range(warmup):
        last = wstep()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = wstep()
    fence()
    dt = time.perf_counter() - t0
    if many:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        all_reduce_max(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    val = float(last.detach())
    grad_bytes = flat.nbytes
    del model, flat, opt
    torch.cuda.empty_cache()
    return {'ms_per_step': round(dt / steps * 1e3, 4), 'samples_per_s': global_batch * steps / dt, 'steps': steps,
            'allreduce_bytes_per_step': {'gradients': grad_bytes, 'loss_scalars': 6 * 4},
            'loss_mean': val,
            'what': 'example SimpleVAE (nn.Linear 784-512-%d / %d-512-784, Adam) fwd + loss + bwd + flat gradient '
                    'all-reduce + optimizer step, global batch %d, B=%d on this GPU, %d GPU(s)'
                    % (2 * L, L, global_batch, B, world)}


def stock_torch_step_rate(logits, noise, x, recon, gz, steps=10):
    import torch
    import torch.distributions as D
    import torch.nn.functional as F
    D.Distribution.set_default_validate_args(False)
    B, L2 = logits.shape
    lg = logits.detach().requires_grad_(True)
    rc = recon.detach().requires_grad_(True)

    def one():
        mu, s = lg[:, :L2 // 2], lg[:, L2 // 2:] + 1e-6
        z = noise * (0.5 * s) + mu
        mu_mean, logvar_mean = mu.mean(), s.mean()
        kld = D.kl_divergence(D.Normal(mu, s), D.Normal(torch.zeros_like(mu), torch.ones_like(s))).sum(-1)
        nll = F.binary_cross_entropy_with_logits(rc, x, reduction='none').view(B, -1).sum(-1)
        elbo, loss = nll + kld, nll + 1.0 * kld
        out = (loss.mean(), elbo.mean(), nll.mean(), kld.mean(), mu_mean, logvar_mean)
        torch.autograd.backward([out[0], z], [None, gz])
        lg.grad = None
        rc.grad = None
        return out
    for _ in range(3):
        one()
    e0, e1 = _Event(), _Event()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        one()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return {'ms_per_step': round(ms, 4), 'samples_per_s': B / (ms * 1e-3),
            'what': 'unfused torch ops + torch.distributions KL + BCE-with-logits + autograd, %d steps' % steps}


def ctypes_off(t, n):
    """Device pointer n elements into tensor t."""
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + n * t.element_size())


if __name__ == '__main__':
    main()
