"""CPU: vae_amd.lazy.LazyParams keeps plain-dict semantics; a marked entry is produced once, on its first read,
whatever the access path (indexing, get, items/values, unpacking, dict(), update, copy, pickle)."""
import copy
import pickle

from vae_amd.lazy import LazyParams


def make(calls):
    d = LazyParams({'z': 1, 'mu': 2})
    d.set_lazy('logvar', lambda: calls.append(1) or 42)
    return d


def test_keys_do_not_materialise():
    calls = []
    d = make(calls)
    assert list(d) == ['z', 'mu', 'logvar'] and set(d.keys()) == {'z', 'mu', 'logvar'}
    assert len(d) == 3 and 'logvar' in d and d.is_pending('logvar') and not calls
    d['mu_mean'] = 0.5
    assert not calls and d['z'] == 1


def test_every_read_path_materialises_once():
    for read in (lambda d: d['logvar'], lambda d: d.get('logvar'), lambda d: dict(d)['logvar'],
                 lambda d: {**d}['logvar'], lambda d: dict(d.items())['logvar'], lambda d: list(d.values())[2],
                 lambda d: d.copy()['logvar'], lambda d: copy.deepcopy(d)['logvar'],
                 lambda d: pickle.loads(pickle.dumps(d))['logvar'], lambda d: (lambda u: u.update(d) or u)({})['logvar'],
                 lambda d: d.pop('logvar'), lambda d: d.setdefault('logvar', 7)):
        calls = []
        d = make(calls)
        assert read(d) == 42
        if 'logvar' in d:
            assert d['logvar'] == 42 and not d.is_pending('logvar')
        assert len(calls) == 1


def test_plain_copies_are_plain_dicts():
    d = make([])
    assert type({**d}) is dict and type(d.copy()) is dict and type(copy.deepcopy(d)) is dict
    assert d == {'z': 1, 'mu': 2, 'logvar': 42}
    assert d.get('absent', 3) == 3
