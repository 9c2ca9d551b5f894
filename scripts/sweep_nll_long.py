"""Cold-cache timing of the long-row NLL kernels (3x64x64 rows, BASELINE config 4)."""
import os, sys
os.environ.setdefault('BENCH_COLD_SAMPLES', '9')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as bc
bc.nll(8192, 12288, 1, 'gaussian')
bc.nll_fused(8192, 12288, 1, 'gaussian')
bc.nll(8192, 12288, 2, 'disc_logistic')
bc.nll_fused(8192, 12288, 2, 'disc_logistic')
bc.nll(8192, 12288, 2, 'disc_logistic', dtype=torch.bfloat16)
bc.nll(2048, 3 * 128 * 128, 1, 'gaussian')
