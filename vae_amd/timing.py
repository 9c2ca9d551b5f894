"""HIP events for kernel timing that do not put a system-scope fence into the stream.

A default HIP event (torch.cuda.Event too) records with a system-scope release: the command processor flushes the
caches to host-visible memory before it takes the timestamp, ~2 us per event on MI355X (measured on a 100 MB launch:
25.5 us between two default events, 23.4 us between two hipEventDisableSystemFence events, 23.0 us by rocprofv3's kernel
trace -- scripts/event_flags_test.py).  For a duration between two points of ONE stream the fence buys nothing, so the
measurement scripts and bench.py's per-kernel durations use events created with hipEventDisableSystemFence; the timed
regions of bench.py (`value`) are host wall-clock between synchronisations and do not depend on this.
"""
import ctypes

import torch

_hip = None
DISABLE_SYSTEM_FENCE = 0x20000000     # hipEventDisableSystemFence (hip_runtime_api.h)


def _lib():
    global _hip
    if _hip is None:
        last = None
        for name in ('libamdhip64.so', 'libamdhip64.so.7', 'libamdhip64.so.6'):
            try:
                _hip = ctypes.CDLL(name)      # the runtime torch has already loaded
                break
            except OSError as e:
                last = e
        if _hip is None:
            raise RuntimeError("vae_amd.timing: the HIP runtime library cannot be loaded: %s" % (last,))
        _hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        _hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        _hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        _hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
    return _hip


class Event(object):
    """record() on torch's current stream (or a given hipStream_t); elapsed_time(other) in milliseconds, like
    torch.cuda.Event."""

    def __init__(self, flags=DISABLE_SYSTEM_FENCE):
        self._e = ctypes.c_void_p()
        rc = _lib().hipEventCreateWithFlags(ctypes.byref(self._e), flags)
        if rc != 0:
            raise RuntimeError("hipEventCreateWithFlags failed (%d)" % rc)

    def record(self, stream=None):
        if stream is None:
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = _lib().hipEventRecord(self._e, stream)
        if rc != 0:
            raise RuntimeError("hipEventRecord failed (%d)" % rc)

    def synchronize(self):
        _lib().hipEventSynchronize(self._e)

    def elapsed_time(self, end):
        ms = ctypes.c_float()
        rc = _lib().hipEventElapsedTime(ctypes.byref(ms), self._e, end._e)
        if rc != 0:
            raise RuntimeError("hipEventElapsedTime failed (%d): both events recorded and complete?" % rc)
        return ms.value

    def __del__(self):
        try:
            if self._e and _hip is not None:
                _hip.hipEventDestroy(self._e)
        except Exception:
            pass
