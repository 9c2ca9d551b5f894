import ctypes, os, sys
sys.path.insert(0, 'scripts')
import torch
import bench_configs as bc
from bench_configs import T, call, ptr
hip = None
for name in ('libamdhip64.so', 'libamdhip64.so.7', 'libamdhip64.so.6'):
    try:
        hip = ctypes.CDLL(name); break
    except OSError:
        pass
print('hip', hip)
def mk(flags):
    e = ctypes.c_void_p()
    rc = hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags))
    assert rc == 0, rc
    return e
st = bc.st
B, L = 65536, 128
logits, z, kl = T(B, 2 * L), T(B, L, kind='e'), T(B, kind='e')
means = torch.zeros(2, device='cuda')
fn = lambda: call('vae_gauss_reparam_kl_fwd', ptr(logits), 2 * L, None, ptr(z), 0, None, ptr(kl), ptr(means), ptr(bc.ws), B, L, 1e-6, 1.0 / (B * L), 1, 1, 1, 0, 0, st)
flush = torch.empty(512 << 18, device='cuda')
for label, flags in (('torch', None), ('hip default', 0), ('hip ReleaseToDevice', 0x40000000), ('hip DisableSystemFence', 0x20000000), ('hip ReleaseToSystem', 0x80000000)):
    samples = []
    for _ in range(21):
        flush.fill_(1.0)
        if flags is None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1))
        else:
            e0, e1 = mk(flags), mk(flags)
            hip.hipEventRecord(e0, st); fn(); hip.hipEventRecord(e1, st); torch.cuda.synchronize()
            ms = ctypes.c_float()
            rc = hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1); assert rc == 0, rc
            samples.append(ms.value)
            hip.hipEventDestroy(e0); hip.hipEventDestroy(e1)
    samples.sort()
    print('%-26s median %.2f us  min %.2f us' % (label, samples[10] * 1e3, samples[0] * 1e3))
