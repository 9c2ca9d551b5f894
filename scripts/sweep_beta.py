"""Cold-cache sweep of the Beta forward (arithmetic-bound): grid cap of the reducing launch, G / U."""
import os, sys
os.environ.setdefault('BENCH_COLD_SAMPLES', '11')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
from vae_amd import _lib
lib = _lib.load()
for mg, G, U in ((0, 0, 0), (0, 16, 2), (2048, 16, 2), (0, 16, 1), (0, 16, 4), (0, 8, 2), (0, 8, 4), (2048, 8, 2), (0, 4, 4), (0, 32, 4), (0, 0, 0)):
    lib.vae_tuning_set(b'MAXGRID', mg); lib.vae_tuning_set(b'G', G); lib.vae_tuning_set(b'U', U)
    print('--- MAXGRID=%d G=%d U=%d' % (mg, G, U), flush=True)
    bc.beta(65536, 512)
lib.vae_tuning_set(b'MAXGRID', 0); lib.vae_tuning_set(b'G', 0); lib.vae_tuning_set(b'U', 0)
