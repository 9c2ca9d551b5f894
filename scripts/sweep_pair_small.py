"""Cold-cache geometry sweep of the per-step VRNN pair launch at step-sized batches (C5: B = 1024..4096, L = 128), in the
form the module path launches it: in-kernel Philox noise, no logvars, no means."""
import os, sys
os.environ.setdefault('BENCH_COLD_SAMPLES', '15')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as bc
from bench_configs import T, call, ptr, timeit
from vae_amd import _lib
lib = _lib.load()


def pair(B, L):
    enc, pri = T(B, 2 * L), T(B, 2 * L)
    zq, zp, cl = T(B, L, kind='e'), T(B, L, kind='e'), T(B, 2 * L, kind='e')
    klp, kl0, gk = T(B, kind='e'), T(B, kind='e'), T(B)
    ge, gp = T(B, 2 * L, kind='e'), T(B, 2 * L, kind='e')
    cfg = 'B=%d L=%d f32 philox, no logvars, no means' % (B, L)
    st, ws = bc.st, bc.ws
    timeit('gauss_pair_fwd', cfg, lambda: call('vae_gauss_pair_fwd', ptr(enc), 2 * L, ptr(pri), 2 * L, None, None, ptr(zq), ptr(zp), None, None, ptr(cl), ptr(klp), ptr(kl0), None, ptr(ws), B, L, 1e-6, 1.0 / (B * L), 1, 7, 3, 0, 0, st),
           B * (32 * L + 8))
    timeit('gauss_pair_bwd', cfg, lambda: call('vae_gauss_pair_bwd', ptr(enc), 2 * L, ptr(pri), 2 * L, None, None, ptr(zq), ptr(zp), None, None, None, ptr(gk), ptr(gk), None, ptr(ge), ptr(gp), B, L, 1e-6, 1.0 / (B * L), 1, 7, 3, 0, 0, st),
           B * (40 * L + 8))


for G, U in ((0, 0), (8, 1), (8, 2), (16, 1), (16, 2), (32, 1), (64, 1)):
    lib.vae_tuning_set(b'G', G); lib.vae_tuning_set(b'U', U)
    print('--- G=%d U=%d' % (G, U), flush=True)
    pair(4096, 128)
    pair(1024, 128)
lib.vae_tuning_set(b'G', 0); lib.vae_tuning_set(b'U', 0)
