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


def test_first_read_under_no_grad_keeps_the_graph():
    """A logging hook that reads the entry under torch.no_grad() must not leave a detached tensor behind: the
    reference builds these entries inside the forward graph (isotropic_gaussian.py:124-125)."""
    import torch
    logits = torch.randn(3, 4, requires_grad=True)
    d = LazyParams({'mu': logits[:, :2]})
    d.set_lazy('logvar', lambda: logits[:, 2:] + 1e-6)
    with torch.no_grad():
        first = d['logvar']
    assert first.requires_grad and d['logvar'] is first
    (g,) = torch.autograd.grad(d['logvar'].sum(), logits)
    assert float(g[:, 2:].sum()) == 6.0 and float(g[:, :2].abs().sum()) == 0.0


def test_eps_tol_follows_config():
    """isotropic_gaussian.py:125 adds helpers.utils.eps(config['half']) (un-vendored: both values assumed, 'parity
    unpinned'); config['eps_tol'] overrides."""
    from vae_amd.reparameterizers.isotropic_gaussian import IsotropicGaussian, EPS_TOL, EPS_TOL_HALF
    base = {'continuous_size': 8, 'cuda': False}
    assert IsotropicGaussian(dict(base, half=False)).eps_tol == EPS_TOL == 1e-6
    assert IsotropicGaussian(dict(base, half=True)).eps_tol == EPS_TOL_HALF == 1e-4
    assert IsotropicGaussian(dict(base, half=True, eps_tol=3e-5)).eps_tol == 3e-5
