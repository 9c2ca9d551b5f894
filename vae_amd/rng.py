"""Counter-based noise bookkeeping for the in-kernel Philox draws.

The reference draws from torch's global generator (isotropic_gaussian.py:88, gumbel.py:269,
torch/distributions/relaxed_bernoulli.py:105).  The kernels instead key Philox4x32-10 by
(seed, call offset, global row, column): `seed` is torch's global seed, `offset` a per-process
call counter that restarts whenever the seed changes, so `torch.manual_seed(s)` still makes a
run reproducible, and a sample's noise does not depend on how the batch is sharded over ranks.
Draws recorded into a HIP graph read their offset from a device word instead (_captured_offset).
For parity tests noise is injected explicitly and none of this is used.
"""
import torch

_state = {'seed': None, 'offset': 0}
OFFSET_ON_DEVICE = 1 << 63      # VAE_OFFSET_ON_DEVICE (include/vae_amd.h)
_cells = []                     # the device words of captured draws: alive as long as the graphs that read them
_graph_devices = set()


def _captured_offset(n):
    """Under HIP graph capture a host (seed, offset) would be frozen into the graph and every replay would repeat
    the draw.  The offset then lives in a device word written by a one-thread node recorded in front of the drawing
    kernel (vae_rng_next: a per-GPU counter inside the library), so every replay draws fresh noise and the backward
    kernel, handed the same word, regenerates it."""
    import ctypes
    from ._lib import call, stream_ptr
    dev = torch.device('cuda', torch.cuda.current_device())
    cell = torch.empty(1, dtype=torch.int64, device=dev)
    _cells.append(cell)
    _graph_devices.add(dev.index)
    call('vae_rng_next', ctypes.c_void_p(cell.data_ptr()), int(n), stream_ptr(dev))
    return OFFSET_ON_DEVICE | cell.data_ptr()


def next_calls(n):
    """(seed, first offset) for n consecutive draws: draw s uses `offset + s`, exactly what n next_call()s give."""
    seed = torch.initial_seed()
    if capturing():
        return seed, _captured_offset(n)
    if seed != _state['seed']:
        _state['seed'] = seed
        _state['offset'] = 0
    first = _state['offset'] + 1
    _state['offset'] += n
    return seed, first


def next_call():
    """(seed, offset) for one kernel call that draws noise."""
    return next_calls(1)


def manual_seed(seed):
    """torch.manual_seed + restart of the call counters (re-seeding torch with the SAME value cannot be
    observed from here, so a script that replays a run inside one process calls this instead)."""
    torch.manual_seed(seed)
    _state['seed'] = torch.initial_seed()
    _state['offset'] = 0
    if _graph_devices:
        from ._lib import call, stream_ptr
        for index in sorted(_graph_devices):
            with torch.cuda.device(index):
                call('vae_rng_reset', stream_ptr(torch.device('cuda', index)))


def capturing():
    """True while torch's current HIP stream is being captured into a graph."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
