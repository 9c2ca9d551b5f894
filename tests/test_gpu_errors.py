"""Error behaviour at the boundary: the C ABI returns status codes (no exceptions cross it), the binding turns
them into RuntimeError; shape / dtype mistakes are caught on the host before any launch (a kernel is never
started on operands whose shapes do not match what its grid assumes).  The reference's own conventions are kept
where it has them: AssertionError for shape asserts (isotropic_gaussian.py:113-120) and unknown registry keys
(reparameterizers/__init__.py:32)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_dtype_and_device_are_checked():
    from vae_amd import functional as F_
    with pytest.raises(TypeError):
        F_.gauss_reparam_kl(torch.randn(4, 8, device=DEV, dtype=torch.float64), None)
    with pytest.raises(TypeError):
        F_.gumbel_softmax(torch.randn(4, 8, device=DEV, dtype=torch.float64))
    with pytest.raises(RuntimeError):
        F_.gauss_kl(torch.randn(4, 8, device=DEV), torch.randn(4, 8))          # mixed devices


def test_shape_asserts_follow_the_reference():
    from vae_amd.reparameterizers import IsotropicGaussian, get_reparameterizer
    m = IsotropicGaussian({'continuous_size': 16, 'cuda': True, 'half': False})
    with pytest.raises(AssertionError):
        m(torch.randn(4, 10, device=DEV))                 # feature size != 2 * output_size
    with pytest.raises(AssertionError):
        m(torch.randn(16, device=DEV))                    # unknown number of dims
    with pytest.raises(AssertionError):
        get_reparameterizer('no_such_reparameterizer')
    with pytest.raises(AssertionError):
        IsotropicGaussian({'continuous_size': 15})


def test_nll_shape_and_type_errors():
    from vae_amd import functional as F_
    x = torch.rand(4, 1, 8, 8, device=DEV)
    with pytest.raises(ValueError):
        F_.nll(x, torch.randn(4, 1, 8, 8, device=DEV), 'gaussian')      # variance NLL needs 2x the channels
    with pytest.raises(ValueError):
        F_.nll(x, torch.randn(4, 2, 8, 8, device=DEV), 'bernoulli')
    with pytest.raises(KeyError):
        F_.nll(x, torch.randn(4, 1, 8, 8, device=DEV), 'pixel_wise')    # not one of the three NLLs of this path


def test_limits_are_status_codes_not_faults():
    from vae_amd import functional as F_, _lib
    with pytest.raises(RuntimeError, match='invalid argument'):
        F_.gumbel_softmax(torch.randn(2, 8, device=DEV), tau=0.0)
    lib = _lib.load()
    st = _lib.stream_ptr(torch.device(DEV))
    t = torch.zeros(8, device=DEV)
    P = lambda v: ctypes.c_void_p(v.data_ptr())
    # null output, negative batch, unknown dtype code, unknown NLL code: VAE_ERR_ARG, nothing launched
    assert lib.vae_gauss_kl_fwd(P(t), 4, P(t), 4, None, 0, None, 0, None, 2, 4, 0, st) == _lib.VAE_ERR_ARG
    assert lib.vae_gauss_kl_fwd(P(t), 4, P(t), 4, None, 0, None, 0, P(t), -1, 4, 0, st) == _lib.VAE_ERR_ARG
    assert lib.vae_gauss_kl_fwd(P(t), 4, P(t), 4, None, 0, None, 0, P(t), 2, 4, 7, st) == _lib.VAE_ERR_ARG
    assert lib.vae_nll_elbo_fwd(P(t), P(t), 9, None, None, None, 1.0, 1.0, P(t), None, None, None, None, 2, 4, 0,
                                st) == _lib.VAE_ERR_ARG
    # B == 0 is a no-op that returns VAE_OK
    assert lib.vae_gauss_kl_fwd(None, 4, None, 4, None, 0, None, 0, None, 0, 4, 0, st) == _lib.VAE_OK
    torch.cuda.synchronize()


def test_unaligned_and_strided_views_take_the_scalar_path():
    """Odd widths and views that break 16-byte alignment must still be right (element-wise kernels)."""
    from vae_amd import functional as F_
    from oracle import cpu_ref as O
    torch.manual_seed(0)
    full = torch.randn(9, 2 * 7 + 3)
    noise = torch.randn(9, 7)
    z0, p0 = O.gauss_forward(full[:, 1:15], noise)
    fd = full.to(DEV)
    z1, lv1, kl1, _ = F_.gauss_reparam_kl(fd[:, 1:15], noise.to(DEV))
    assert torch.equal(z1.cpu(), z0)
    torch.testing.assert_close(kl1.cpu(), O.gauss_kl(p0), rtol=1e-5, atol=1e-4)
