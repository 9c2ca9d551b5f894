"""The example host model (examples/train_simple_vae.py) trains: the loss after 40 steps is well below the first."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_trains():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'train_simple_vae.py'), '--steps', '41',
                          '--batch', '1024'], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    losses = [float(m) for m in re.findall(r'loss ([0-9.]+)', out.stdout)]
    assert len(losses) >= 4 and losses[-1] < 0.7 * losses[0], out.stdout


def test_example_trains_two_ranks_gloo():
    """The DDP wiring of the example (vae_amd.dist.configure + globalize_loss + DistributedDataParallel) with two
    ranks sharing the GPU over gloo: the global-batch loss falls."""
    env = dict(os.environ, EXAMPLE_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', '29533',
                          os.path.join(ROOT, 'examples', 'train_simple_vae.py'), '--steps', '41', '--batch', '512'],
                         capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    losses = [float(m) for m in re.findall(r'loss ([0-9.]+)', out.stdout)]
    assert len(losses) >= 4 and losses[-1] < 0.7 * losses[0], out.stdout
