"""The C ABI without torch: tests/abi_client/elbo_step_client (plain C + the HIP runtime API + dlopen, built by
__graft_entry__.build()) runs one ELBO step through libvae_amd.so and checks it against a scalar restatement of
the reference's arithmetic."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_runs_the_step():
    exe = os.path.join(ROOT, 'tests', 'abi_client', 'elbo_step_client')
    lib = os.path.join(ROOT, 'vae_amd', 'libvae_amd.so')
    assert os.path.exists(exe), 'tests/abi_client/elbo_step_client is missing: run __graft_entry__.build()'
    out = subprocess.run([exe, lib], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith('OK'), out.stdout
