"""Cold-cache (G, U) sweep of the fused VRNN pair launch at bandwidth-bound sizes (register-heavy: 168-251 VGPRs at U=2)."""
import os, sys
os.environ.setdefault('BENCH_COLD_SAMPLES', '11')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
from vae_amd import _lib
lib = _lib.load()
for G, U in ((0, 0), (0, 1), (32, 1), (64, 1), (32, 2), (0, 0)):
    lib.vae_tuning_set(b'G', G); lib.vae_tuning_set(b'U', U)
    print('--- G=%d U=%d' % (G, U), flush=True)
    bc.pair_fused(65536, 512)
    bc.pair_fused(65536, 128)
lib.vae_tuning_set(b'G', 0); lib.vae_tuning_set(b'U', 0)
