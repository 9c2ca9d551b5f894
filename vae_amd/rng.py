"""Counter-based noise bookkeeping for the in-kernel Philox draws.

The reference draws from torch's global generator (isotropic_gaussian.py:88, gumbel.py:269,
torch/distributions/relaxed_bernoulli.py:105).  The kernels instead key Philox4x32-10 by
(seed, call offset, global row, column): `seed` is torch's global seed, `offset` a per-process
call counter that restarts whenever the seed changes, so `torch.manual_seed(s)` still makes a
run reproducible, and a sample's noise does not depend on how the batch is sharded over ranks.
Draws recorded into a HIP graph read their offset from a device word instead (_captured_offset).
The SEED of a captured draw is a host value recorded into the graph: `manual_seed(new)` after a capture restarts the
device counter (so a replayed run is reproducible) but replays keep the seed they were captured with -- re-capture to
reseed.  For parity tests noise is injected explicitly and none of this is used.
"""
import torch

_state = {'seed': None, 'offset': 0}
OFFSET_ON_DEVICE = 1 << 63      # VAE_OFFSET_ON_DEVICE (include/vae_amd.h)
_POOL_WORDS = 1 << 16           # captured draws per device between two release_captured() calls (512 KiB)
_pools = {}                     # device index -> [int64 tensor allocated OUTSIDE any capture, next free word]
_cells = []                     # fallback words allocated under capture (no eager call came first on that device)
_graph_devices = set()


def _pool(dev, create):
    p = _pools.get(dev.index)
    if p is None and create:
        p = _pools[dev.index] = [torch.zeros(_POOL_WORDS, dtype=torch.int64, device=dev), 0]
    return p


def release_captured():
    """Hand back the device words of every captured draw.  Call it after the graphs that were captured so far have
    been deleted (a model that re-captures on every shape change would otherwise run out of the 65536 words a device
    holds); graphs that are still alive would share words with later captures."""
    for p in _pools.values():
        p[1] = 0
    del _cells[:]


def _captured_offset(n):
    """Under HIP graph capture a host (seed, offset) would be frozen into the graph and every replay would repeat
    the draw.  The offset then lives in a device word written by a one-thread node recorded in front of the drawing
    kernel (vae_rng_next: a per-GPU counter inside the library), so every replay draws fresh noise and the backward
    kernel, handed the same word, regenerates it."""
    import ctypes
    from ._lib import call, stream_ptr
    dev = torch.device('cuda', torch.cuda.current_device())
    p = _pool(dev, create=False)
    if p is None:
        # no eager draw ran on this device before the capture (torch asks for warm-up steps, which create the pool):
        # a word from the graph's own memory pool, kept alive here
        cell = torch.empty(1, dtype=torch.int64, device=dev)
        _cells.append(cell)
        address = cell.data_ptr()
    else:
        if p[1] >= _POOL_WORDS:
            raise RuntimeError("vae_amd.rng: %d captured draws on cuda:%d; delete the old graphs and call "
                               "vae_amd.rng.release_captured()" % (_POOL_WORDS, dev.index))
        address = p[0].data_ptr() + 8 * p[1]
        p[1] += 1
    _graph_devices.add(dev.index)
    call('vae_rng_next', ctypes.c_void_p(address), int(n), stream_ptr(dev))
    return OFFSET_ON_DEVICE | address


def next_calls(n):
    """(seed, first offset) for n consecutive draws: draw s uses `offset + s`, exactly what n next_call()s give."""
    seed = torch.initial_seed()
    if capturing():
        return seed, _captured_offset(n)
    if seed != _state['seed']:
        _state['seed'] = seed
        _state['offset'] = 0
    if not _pools and torch.cuda.is_available():
        # outside any capture; once per process (one process per GPU): a second device used by the same process falls
        # back to words from its graph's own pool (_captured_offset)
        _pool(torch.device('cuda', torch.cuda.current_device()), create=True)
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
