"""A params dict whose marked entries are produced on first read.

The reference materialises `params['gaussian']['logvar'] = logits[..., L:] + eps` on every forward
(isotropic_gaussian.py:124-125), a [B, L] tensor that the training step itself never reads again: the KL and the
sample come out of the fused kernel.  `LazyParams` keeps the reference's dict schema -- same keys, and any read of
the entry (indexing, get, items, values, copying, unpacking, pickling) returns exactly the tensor the reference
would hold, computed by the very op the reference uses -- but the 4L bytes per sample are only written if somebody
asks for them.
"""


import torch


class _Pending(object):
    __slots__ = ('fn',)

    def __init__(self, fn):
        self.fn = fn


class LazyParams(dict):
    def set_lazy(self, key, fn):
        """`fn()` -> the entry's value, called once, on the first read."""
        dict.__setitem__(self, key, _Pending(fn))

    def is_pending(self, key):
        return isinstance(dict.get(self, key), _Pending)

    def _resolve(self, key, value):
        if isinstance(value, _Pending):
            # the reference builds these entries inside the forward graph: a first read under torch.no_grad() (a
            # logging hook, say) must not cache a detached tensor that a later kl() / mutual_info() would then
            # differentiate through in vain
            with torch.enable_grad():
                value = value.fn()
            dict.__setitem__(self, key, value)
        return value

    def __getitem__(self, key):
        return self._resolve(key, dict.__getitem__(self, key))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __iter__(self):
        # a Python-level __iter__ also takes dict(d), {**d} and d2.update(d) off CPython's raw-storage fast path:
        # they then read every entry through __getitem__
        return iter(self.keys())

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def pop(self, key, *default):
        if key in self:
            value = self[key]
            dict.__delitem__(self, key)
            return value
        return dict.pop(self, key, *default)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        dict.__setitem__(self, key, default)
        return default

    def copy(self):
        return dict(self.items())

    def __eq__(self, other):
        return dict(self.items()) == other

    __hash__ = None

    def __reduce__(self):
        return (dict, (dict(self.items()),))     # pickles / deep-copies as a plain dict


def hard_from_soft_max(y):
    """Straight-through one-hot of a soft sample, from the sample alone (gumbel.py:296-301): 1 where y equals its
    row maximum (multi-hot on exact ties), value of the mask, gradient of y.  Plain elementwise torch ops on the
    device: the same IEEE operations the fused kernel performs, so the bits are the same."""
    yd = y.detach()
    h = (yd == yd.amax(-1, keepdim=True)).to(y.dtype)
    return (h - yd) + y


def hard_from_soft_threshold(y):
    """Straight-through 0/1 of a relaxed Bernoulli sample (bernoulli.py:111-124): 1 where y >= 0.5."""
    yd = y.detach()
    h = (yd >= 0.5).to(y.dtype)
    return (h - yd) + y
