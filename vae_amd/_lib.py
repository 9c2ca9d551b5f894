"""ctypes binding of libvae_amd.so (the C ABI declared in include/vae_amd.h).

The library holds only HIP kernels + `extern "C"` launchers (no torch headers);
torch is used here for device memory and the current HIP stream, nothing else.
`import torch` must happen before the library is opened so that both resolve the
same `libamdhip64.so.7` (torch ships its own copy; the soname is shared).

There is NO CPU fallback: a missing library or a CPU tensor raises.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported before the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VAE_AMD_LIB') or os.path.join(_HERE, 'libvae_amd.so')  # env override: tuning builds only
ABI_VERSION = 11

VAE_OK, VAE_ERR_ARG, VAE_ERR_LAUNCH = 0, 1, 2
VAE_F32, VAE_BF16, VAE_F16 = 0, 1, 2
NLL_TYPES = {'bernoulli': 0, 'gaussian': 1, 'disc_logistic': 2, 'log_logistic_256': 2}

_p, _i64, _u64, _f, _i = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float, ctypes.c_int
_pp = ctypes.POINTER(ctypes.c_void_p)   # host array of device pointers
_fp = ctypes.POINTER(ctypes.c_float)    # host array of floats

# name -> argtypes; mirrors include/vae_amd.h one to one (tests/test_abi.py checks the symbols)
SIGNATURES = {
    'vae_abi_version': [],
    'vae_workspace_bytes': [],
    'vae_rng_next': [_p, _i64, _p],
    'vae_rng_reset': [_p],
    'vae_nan_flag': [_p, _i, ctypes.POINTER(ctypes.c_uint32), _p],
    'vae_tuning_set': [ctypes.c_char_p, _i64],
    'vae_gauss_reparam_kl_fwd': [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i64, _i64, _f, _f, _i, _u64, _u64, _i64, _i,
                                 _p],
    'vae_gauss_reparam_kl_bwd': [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i64, _i64, _i64, _f, _f, _i, _u64, _u64, _i64,
                                 _i, _p],
    'vae_gauss_reparam_kl_fwd_nchw': [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _f, _f, _i, _u64, _u64, _i64, _i,
                                      _p],
    'vae_gauss_reparam_kl_bwd_nchw': [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _f, _f, _i, _u64, _u64, _i64,
                                      _i, _p],
    'vae_gauss_kl_fwd': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i, _p],
    'vae_gauss_kl_bwd': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_gauss_pair_fwd': [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _f, _f, _i,
                           _u64, _u64, _i64, _i, _p],
    'vae_gauss_pair_bwd': [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _f, _f,
                           _i, _u64, _u64, _i64, _i, _p],
    'vae_gumbel_fwd': [_p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _f, _u64, _u64, _i64, _i, _p],
    'vae_gumbel_bwd': [_p, _i64, _p, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _f, _i, _p],
    'vae_concat_gauss_gumbel_supported': [_i64, _i64, _i],
    'vae_concat_gauss_gumbel_fwd': [_p, _i64, _p, _p, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _f, _f, _i, _u64, _u64,
                                    _u64, _u64, _i64, _i, _p],
    'vae_concat_gauss_gumbel_bwd': [_p, _i64, _p, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _f, _f, _u64,
                                    _u64, _i64, _i, _p],
    'vae_categorical_kl_fwd': [_p, _i64, _p, _i64, _p, _i64, _i64, _i, _p],
    'vae_categorical_kl_bwd': [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_bernoulli_fwd': [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _f, _u64, _u64, _i64, _i, _p],
    'vae_bernoulli_bwd': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _f, _i, _p],
    'vae_bernoulli_kl_fwd': [_p, _i64, _p, _i64, _p, _i64, _i64, _i, _p],
    'vae_bernoulli_kl_bwd': [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_beta_split_kl_fwd': [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _f, _i, _p],
    'vae_beta_split_kl_bwd': [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _f, _i, _p],
    'vae_beta_kl_fwd': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i, _p],
    'vae_beta_kl_bwd': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_flow_kl_fwd': [_p, _p, _i64, _p, _p, _p, _i64, _i64, _f, _i, _i, _i, _p],
    'vae_flow_kl_bwd': [_p, _p, _p, _p, _i64, _i64, _f, _i, _p],
    'vae_nll_elbo_fwd': [_p, _p, _i, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_nll_elbo_bwd': [_p, _p, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_nll_elbo_fused': [_p, _p, _i, _p, _p, _p, _f, _f, _f, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_nll_fused_grad_finish': [_p, _p, _i64, _i, _p],
    'vae_nll_elbo_fused_static': [_p, _p, _i, _p, _p, _p, _f, _f, _f, _p, _p, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_nll_fused_backward': [_p, _i64, _p, _f, _f, _p, _p, _p, _i64, _i, _p],
    'vae_elbo_combine_fwd': [_p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i64, _i, _p],
    'vae_elbo_combine_bwd': [_p, _p, _p, _f, _f, _p, _p, _p, _p, _i64, _i, _p],
    'vae_gauss_reparam_multi_fwd': [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _f, _u64, _u64, _i64, _i, _p],
    'vae_gauss_reparam_multi_bwd': [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _f, _u64, _u64, _i64, _i,
                                    _p],
    'vae_nll_multi_elbo_fwd': [_p, _pp, _i64, _i, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_nll_multi_elbo_bwd': [_p, _pp, _pp, _i64, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_nll_steps_elbo_fwd': [_pp, _pp, _i64, _i, _p, _p, _p, _fp, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64,
                               _i, _p],
    'vae_nll_steps_elbo_bwd': [_pp, _pp, _pp, _i64, _i, _p, _p, _p, _p, _fp, _f, _f, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_made_tail_fwd': [_p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i, _i, _p],
    'vae_made_tail_moments_supported': [_i64, _i],
    'vae_made_tail_fwd_moments': [_p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i64, _i, _i, _p],
    'vae_made_tail_bwd': [_p, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i, _i, _p],
    'vae_coupling_tail_fwd': [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_coupling_tail_bwd': [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _p],
    'vae_bnflow_stats_floats': [_i64],
    'vae_bnflow_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _f, _i, _i, _i, _p],
    'vae_bnflow_bwd': [_p, _p, _p, _p, _p, _i64, _i64, _i, _i, _i, _p],
    'vae_bnflow_bwd_sums': [_p, _p, _p, _p, _i64, _i64, _i, _i, _p],
    'vae_bnflow_bwd_apply': [_p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _p],
}

_lib = None


def load():
    """Open libvae_amd.so (once).  Raises if it was not built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "vae_amd: %s is missing; build it with `make -C vae_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU/eager fallback for this path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.argtypes = argtypes
        fn.restype = ctypes.c_size_t if name in ('vae_workspace_bytes', 'vae_bnflow_stats_floats') else ctypes.c_int
    if lib.vae_abi_version() != ABI_VERSION:
        raise RuntimeError("vae_amd: libvae_amd.so has ABI %d, the binding expects %d; rebuild"
                           % (lib.vae_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def dtype_code(t):
    if t.dtype == torch.float32:
        return VAE_F32
    if t.dtype == torch.bfloat16:
        return VAE_BF16
    if t.dtype == torch.float16:
        return VAE_F16
    raise TypeError("vae_amd kernels take float32, bfloat16 or float16 tensors, got %s" % t.dtype)


def check_device(*tensors):
    """All given (non-None) tensors must live on one ROCm device; returns it."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("vae_amd: tensor on %s; the HIP path has no CPU fallback "
                               "(use oracle/cpu_ref.py only as a test oracle)" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("vae_amd: tensors on different devices (%s vs %s)" % (dev, t.device))
    return dev


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def float_array(values):
    """Host array of floats (the per-step KL weights of the step-table NLL), or None."""
    return None if values is None else (ctypes.c_float * len(values))(*[float(v) for v in values])


def ptr_array(tensors):
    """Host array of device pointers (the multi-sample NLL takes its S decoder outputs this way)."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_handle(device):
    """The current HIP stream of `device` as an integer (the raw accessor skips building a Stream object: ~8 us
    per call on the module path, which launches 30-40 us kernels)."""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def stream_ptr(device):
    return ctypes.c_void_p(_stream_handle(device))


_workspaces = {}


def workspace(device):
    """Per (device, stream) scratch for block partials; one call at a time per stream."""
    key = (device.index, _stream_handle(device))
    ws = _workspaces.get(key)
    if ws is None:
        nbytes = load().vae_workspace_bytes()
        ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)  # zeroed once: ticket counters
        _workspaces[key] = ws
    return ws


NAN_BITS = {1: 'nll / loss / elbo', 2: 'kld', 4: 'reparameterizer parameters (encoder logits)',
            8: 'additional loss / mutual information / flow KL'}


def nan_flags(device=None, clear=True):
    """OR of the sticky saw-a-NaN words (include/vae_amd.h: VAE_NAN_*) of every workspace this process holds on
    `device` (default: the current one).  Synchronises the streams those workspaces belong to: meant for the end of
    an epoch or every N steps -- the kernels raise the bits without any host involvement (the reference's
    nan_check_and_break, abstract_vae.py:366-369, synchronises on every step instead).  The word of the library's
    static scratch (vae_nll_elbo_fused_static: callers of the C ABI only, nothing in this package launches it) is
    included: vae_nan_flag with workspace = NULL reads that one."""
    index = torch.cuda.current_device() if device is None else torch.device(device).index
    index = torch.cuda.current_device() if index is None else index
    flags = 0
    out = ctypes.c_uint32(0)
    with torch.cuda.device(index):
        for (dev_index, stream), ws in list(_workspaces.items()):
            if dev_index != index:
                continue
            call('vae_nan_flag', ctypes.c_void_p(ws.data_ptr()), int(bool(clear)), ctypes.byref(out),
                 ctypes.c_void_p(stream))
            flags |= out.value
        call('vae_nan_flag', None, int(bool(clear)), ctypes.byref(out), stream_ptr(torch.device('cuda', index)))
        flags |= out.value
    return flags


def describe_nan_flags(flags):
    return ', '.join(name for bit, name in NAN_BITS.items() if flags & bit)


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != VAE_OK:
        what = {VAE_ERR_ARG: 'invalid argument', VAE_ERR_LAUNCH: 'kernel launch failed'}.get(rc, 'error %d' % rc)
        raise RuntimeError("vae_amd: %s: %s" % (name, what))
