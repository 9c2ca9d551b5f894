"""The example host model (examples/train_simple_vae.py) trains: the loss after 40 steps is well below the first."""
import os
import re
import signal
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(cmd, env=None, timeout=100):
    """Run in its own process group; a run that does not finish is killed as a group (no stray ranks left on the GPU)
    and reported, never waited on."""
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        pytest.fail('timed out after %d s: %s\n%s' % (timeout, ' '.join(cmd), err[-2000:]))
    return p.returncode, out, err


def _check_losses(rc, out, err):
    assert rc == 0, err[-3000:]
    losses = [float(m) for m in re.findall(r'loss ([0-9.]+)', out)]
    assert len(losses) >= 4 and losses[-1] < 0.7 * losses[0], out


def test_example_trains():
    _check_losses(*_run([sys.executable, os.path.join(ROOT, 'examples', 'train_simple_vae.py'), '--steps', '41',
                         '--batch', '1024']))


def test_example_trains_two_ranks_gloo():
    """The multi-rank wiring of the example (vae_amd.dist.configure + globalize_loss + summed gradients) with two
    ranks sharing the GPU over gloo: the global-batch loss falls.  (DDP itself needs one GPU per rank over RCCL;
    its gloo reduction of device tensors stalled intermittently with both ranks on one GPU.)"""
    env = dict(os.environ, EXAMPLE_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    _check_losses(*_run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                         os.path.join(ROOT, 'examples', 'train_simple_vae.py'), '--steps', '41', '--batch', '512'],
                        env=env))


def test_bench_multi_rank_path_runs_under_rccl_with_one_rank():
    """bench.py's N > 1 code path -- process group on RCCL, the per-step scalar all-reduce on the collective's stream
    beside the backward launches, FlatGradients / globalize_loss in the whole-model leg, the strong leg's graph replay
    + all-reduce on a copy -- with a single rank (BENCH_FORCE_DIST / VAE_AMD_DIST_FORCE): the one way to execute the
    RCCL side on a one-GPU box.  The loss must equal the single-process step's (same seeds, same kernels)."""
    import json
    env = dict(os.environ, BENCH_FORCE_DIST='1', VAE_AMD_DIST_FORCE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    rc, out, err = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                         os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2', '--batch', '8192',
                         '--no-cpu-baseline', '--no-configs'], env=env, timeout=150)
    assert rc == 0, err[-3000:]
    line = [l for l in out.splitlines() if l.startswith('{')]
    assert len(line) == 1, out[-2000:]
    j = json.loads(line[0])
    assert j['n_gpus'] == 1 and j['scaling'] == 'weak' and j['value'] > 0
    assert j['whole_model']['samples_per_s'] > 0 and j['whole_model']['loss_mean'] == j['whole_model']['loss_mean']
    rc0, out0, err0 = _run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '5', '--warmup', '2', '--batch',
                            '8192', '--main-only'], timeout=100)
    assert rc0 == 0, err0[-3000:]
    j0 = json.loads([l for l in out0.splitlines() if l.startswith('{')][0])
    assert abs(j['loss_mean'] - j0['loss_mean']) <= 1e-6 * abs(j0['loss_mean'])
