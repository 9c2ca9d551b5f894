"""Cold-cache sweep of the launch geometry (lanes per row G, chunks in flight U, grid caps) for single kernels,
through vae_tuning_set: `python scripts/sweep_geom.py` on the GPU box.  Uses scripts/bench_configs.py's timing
(each launch behind a 512 MiB fill, median of 9)."""
import os
import sys

os.environ.setdefault('BENCH_COLD_SAMPLES', '31')

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as bc
from vae_amd import _lib

lib = _lib.load()


def knobs(G=0, U=0, maxgrid=0, reducegrid=0):
    for name, v in (('G', G), ('U', U), ('MAXGRID', maxgrid), ('REDUCEGRID', reducegrid)):
        assert lib.vae_tuning_set(name.encode(), v) == 0


cases = {
    'gauss_bf16': lambda: bc.gauss(65536, 512, dtype=torch.bfloat16),
    'nll_bf16': lambda: bc.nll(65536, 784, 0, 'bernoulli', dtype=torch.bfloat16),
    'gauss_philox': lambda: bc.gauss(65536, 512, philox=True),
    'gauss_f32': lambda: bc.gauss(65536, 512),
    'pair': lambda: bc.pair_fused(16384, 128),
    'pair512': lambda: bc.pair_fused(65536, 512),
    'pair64': lambda: bc.pair_fused(65536, 64),
    'flow64': lambda: bc.flow(262144, 64),
    'flow2048': lambda: bc.flow(16384, 2048),
    'gauss_L128_B16k': lambda: bc.gauss(16384, 128),
    'gumbel256': lambda: bc.gumbel(65536, 256),
    'flow': lambda: bc.flow(65536, 512),
    'bnflow': lambda: bc.flowtail(65536, 512),
    'gauss_f32_L1024': lambda: bc.gauss(65536, 1024),
    'gauss_f32_L256': lambda: bc.gauss(131072, 256),
    'gauss_f32_L128': lambda: bc.gauss(262144, 128),
    'gauss_philox_L1024': lambda: bc.gauss(65536, 1024, philox=True),
    'gauss_philox_L128': lambda: bc.gauss(262144, 128, philox=True),
    'gauss_bf16_L1024': lambda: bc.gauss(65536, 1024, dtype=torch.bfloat16),
    'nll_fused': lambda: bc.nll_fused(65536, 784, 0, 'bernoulli'),
    'nll_f32': lambda: bc.nll(65536, 784, 0, 'bernoulli'),
}
which = sys.argv[1:] or list(cases)
for name in which:
    for G, U in ((0, 0), (16, 1), (16, 2), (16, 4), (32, 1), (32, 2), (32, 4), (64, 1), (64, 2), (64, 4)):
        for rg in (0, 2048):
            knobs(G=G, U=U, reducegrid=rg)
            print('--- %s G=%d U=%d REDUCEGRID=%d' % (name, G, U, rg), flush=True)
            cases[name]()
knobs()
