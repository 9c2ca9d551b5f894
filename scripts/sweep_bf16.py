"""Cold-cache geometry sweep of the bf16 C2 kernels (vae_tuning_set G / U)."""
import os, sys
os.environ.setdefault('BENCH_COLD_SAMPLES', '15')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as bc
from vae_amd import _lib
lib = _lib.load()
for G, U in ((0, 0), (16, 4), (32, 2), (32, 4), (64, 1), (64, 2), (64, 4)):
    lib.vae_tuning_set(b'G', G); lib.vae_tuning_set(b'U', U)
    print('--- G=%d U=%d' % (G, U), flush=True)
    bc.nll_fused(65536, 784, 0, 'bernoulli', dtype=torch.bfloat16)
    bc.gauss(65536, 512, dtype=torch.bfloat16)
lib.vae_tuning_set(b'G', 0); lib.vae_tuning_set(b'U', 0)
