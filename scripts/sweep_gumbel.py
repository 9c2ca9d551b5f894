"""Cold-cache sweep of the lane-group width of the register-resident softmax kernels (gumbel forward / backward,
categorical KL) through vae_tuning_set("G"): `python scripts/sweep_gumbel.py [K ...]` on the GPU box."""
import os
import sys

os.environ.setdefault('BENCH_COLD_SAMPLES', '15')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
from vae_amd import _lib

lib = _lib.load()
Ks = [int(a) for a in sys.argv[1:]] or [256]
for K in Ks:
    for G in (0, 16, 32, 64):
        assert lib.vae_tuning_set(b'G', G) == 0
        print('--- K=%d G=%d' % (K, G), flush=True)
        bc.gumbel(65536, K, philox=True)
        bc.gumbel(65536, K)
lib.vae_tuning_set(b'G', 0)
