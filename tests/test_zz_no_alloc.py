"""The allocation-free entry point (vae_nll_elbo_fused_static: its scratch is a __device__ array of the library)
against vae_nll_elbo_fused with a caller workspace, through the C ABI: every output bit for bit (same kernels, same
summation order), on two calls in a row, the second with a smaller input -- the array keeps the first call's
contents."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(name, x, recon, kld, nll_type, dtype_code, with_ws):
    from vae_amd import _lib
    from vae_amd._lib import call, ptr
    B, D = x.shape
    rw = recon.shape[1]
    mk = lambda *s: torch.full(s, float('nan'), dtype=x.dtype, device=DEV)
    nll, loss, elbo, grecon = mk(B), mk(B), mk(B), mk(B, rw)
    means = torch.full((6,), float('nan'), device=DEV)
    st = _lib.stream_ptr(torch.device(DEV))
    args = [ptr(x), ptr(recon), nll_type, ptr(kld), None, None, 0.7, 1.0 / B, 1.0, ptr(nll), ptr(loss), ptr(elbo),
            ptr(means)]
    if with_ws:
        args.append(ptr(_lib.workspace(torch.device(DEV))))
    call(name, *args, ptr(grecon), B, D, dtype_code, st)
    torch.cuda.synchronize()
    return nll, loss, elbo, means, grecon


@pytest.mark.parametrize('nll_type,dtype', [(0, torch.float32), (2, torch.float32), (0, torch.bfloat16)])
def test_static_scratch_matches_workspace_entry_point(nll_type, dtype):
    code = {torch.float32: 0, torch.bfloat16: 1}[dtype]
    g = torch.Generator(device=DEV).manual_seed(7 + nll_type)
    # a big call first, then smaller ones: stale partials / counters of the first must not leak into the next.
    # The first two sizes force the MULTI-BLOCK closing launch (more than 8 * 256 vectors of per-sample values:
    # B > 8192 in fp32, > 16384 in the 16-bit types), i.e. the path that really uses the __device__ scratch -- the
    # arrival counter, the block partials, the address from hipGetSymbolAddress; the last two close in one block.
    big = (70001, 9001) if dtype == torch.float32 else (70001, 17001)
    for B, D in ((big[0], 784), (big[1], 784), (37, 784), (1, 16)):
        x = torch.rand(B, D, generator=g, device=DEV).to(dtype)
        recon = torch.randn(B, D if nll_type == 0 else 2 * D, generator=g, device=DEV).to(dtype)
        kld = torch.rand(B, generator=g, device=DEV).to(dtype)
        ref = _run('vae_nll_elbo_fused', x, recon, kld, nll_type, code, True)
        got = _run('vae_nll_elbo_fused_static', x, recon, kld, nll_type, code, False)
        for r, o, what in zip(ref, got, ('nll', 'loss', 'elbo', 'means', 'grecon')):
            assert not bool(torch.isnan(o.float()).any()), what
            assert torch.equal(r, o), (what, B, D)


@pytest.mark.parametrize('nll_type', [0, 2])
def test_static_scratch_long_rows_to_rounding(nll_type):
    """Rows of more than 1024 vectors: the workspace form sums a row in chunks (one block per chunk, partial sums in
    the workspace), this one a block per row -- the gradient bit for bit, nll / loss / elbo and the means to the
    rounding of another summation order (include/vae_amd.h says so)."""
    g = torch.Generator(device=DEV).manual_seed(5)
    for B, D in ((300, 3 * 64 * 64), (33, 5000)):
        x = torch.rand(B, D, generator=g, device=DEV)
        recon = torch.randn(B, D if nll_type == 0 else 2 * D, generator=g, device=DEV)
        kld = torch.rand(B, generator=g, device=DEV)
        ref = _run('vae_nll_elbo_fused', x, recon, kld, nll_type, 0, True)
        got = _run('vae_nll_elbo_fused_static', x, recon, kld, nll_type, 0, False)
        assert torch.equal(ref[4], got[4])
        for r, o in zip(ref[:4], got[:4]):
            torch.testing.assert_close(o, r, rtol=2e-6, atol=0)


def test_static_scratch_nan_word():
    """A NaN sample in a multi-block call raises the sticky word of the static scratch (read with workspace = NULL),
    and only there; clearing it works."""
    from vae_amd import _lib
    from vae_amd._lib import call
    g = torch.Generator(device=DEV).manual_seed(3)
    B, D = 20001, 64
    x = torch.rand(B, D, generator=g, device=DEV)
    recon = torch.randn(B, D, generator=g, device=DEV)
    kld = torch.rand(B, generator=g, device=DEV)
    st = _lib.stream_ptr(torch.device(DEV))
    out = ctypes.c_uint32(99)
    call('vae_nan_flag', None, 1, ctypes.byref(out), st)            # start clean
    _run('vae_nll_elbo_fused_static', x, recon, kld, 0, 0, False)
    call('vae_nan_flag', None, 0, ctypes.byref(out), st)
    assert out.value == 0
    recon[12345, 7] = float('nan')
    got = _run('vae_nll_elbo_fused_static', x, recon, kld, 0, 0, False)
    assert bool(torch.isnan(got[3][0]))                             # loss_mean
    call('vae_nan_flag', None, 1, ctypes.byref(out), st)
    assert out.value & 1                                            # VAE_NAN_NLL
    call('vae_nan_flag', None, 0, ctypes.byref(out), st)
    assert out.value == 0                                           # cleared by the previous read


def test_static_scratch_rejects_oversize():
    from vae_amd import _lib
    lib = _lib.load()
    x = torch.zeros(4, 8, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.vae_nll_elbo_fused_static(p(x), p(x), 0, None, None, None, 1.0, 1.0, 1.0, p(x), p(x), p(x), p(x), p(x),
                                       (1 << 24) + 1, 8, 0, None)
    assert rc == _lib.VAE_ERR_ARG
