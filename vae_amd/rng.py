"""Counter-based noise bookkeeping for the in-kernel Philox draws.

The reference draws from torch's global generator (isotropic_gaussian.py:88, gumbel.py:269,
torch/distributions/relaxed_bernoulli.py:105).  The kernels instead key Philox4x32-10 by
(seed, call offset, global row, column): `seed` is torch's global seed, `offset` a per-process
call counter that restarts whenever the seed changes, so `torch.manual_seed(s)` still makes a
run reproducible, and a sample's noise does not depend on how the batch is sharded over ranks.
For parity tests noise is injected explicitly and none of this is used.
"""
import torch

_state = {'seed': None, 'offset': 0}


def next_call():
    """(seed, offset) for one kernel call that draws noise."""
    seed = torch.initial_seed()
    if seed != _state['seed']:
        _state['seed'] = seed
        _state['offset'] = 0
    _state['offset'] += 1
    return seed, _state['offset']


def next_calls(n):
    """(seed, first offset) for n consecutive draws: draw s uses `offset + s`, exactly what n next_call()s give."""
    seed, first = next_call()
    _state['offset'] += n - 1
    return seed, first


def manual_seed(seed):
    """torch.manual_seed + restart of the call counter (re-seeding torch with the SAME value cannot be
    observed from here, so a script that replays a run inside one process calls this instead)."""
    torch.manual_seed(seed)
    _state['seed'] = torch.initial_seed()
    _state['offset'] = 0


def capturing():
    """True while torch's current HIP stream is being captured into a graph."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def captured_noise(kind, shape, like):
    """Noise for a draw issued during HIP graph capture, else None.  The (seed, offset) pair of an in-kernel
    Philox draw is a host value: captured, it would be frozen into the graph and every replay would repeat the
    same draw.  torch's generator is graph-aware (its Philox offset lives on the device and advances on replay),
    so while capturing the draw is made by torch and handed to the kernel as injected noise."""
    if not capturing():
        return None
    if kind == 'normal':
        return torch.randn(shape, dtype=like.dtype, device=like.device)
    return torch.rand(shape, dtype=like.dtype, device=like.device)
