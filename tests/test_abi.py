"""CPU: libvae_amd.so loads and exports every symbol include/vae_amd.h declares; the ctypes table
in vae_amd/_lib.py lists exactly those symbols with the header's arity.  No compute calls."""
import os
import re
import ctypes

from vae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    src = open(os.path.join(ROOT, 'include', 'vae_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(?:int|size_t)\s+(vae_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ('', 'void') else len(args.split(','))
    return decls


def test_library_exports_header_symbols():
    decls = header_decls()
    assert len(decls) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), name


def test_binding_matches_header():
    decls = header_decls()
    assert set(decls) == set(_lib.SIGNATURES)
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name]) == n, name


def test_load_and_version():
    lib = _lib.load()
    assert lib.vae_abi_version() == _lib.ABI_VERSION
    assert lib.vae_workspace_bytes() >= 2048 * 8 * 4


def test_cpu_tensor_is_rejected():
    import torch
    import pytest
    from vae_amd import functional as F_
    with pytest.raises(RuntimeError):
        F_.gauss_reparam_kl(torch.randn(4, 8), torch.randn(4, 4))
    with pytest.raises(RuntimeError):
        F_.nll(torch.rand(2, 1, 4, 4), torch.randn(2, 1, 4, 4), 'bernoulli')
