import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import torch
    gdir = os.path.join(ROOT, 'tests', 'golden')
    cache = {}

    def load(name):
        if name not in cache:
            # fixtures were written by tests/golden/make_golden.py in this repo: plain tensors / floats / dicts
            cache[name] = torch.load(os.path.join(gdir, name + '.pt'), weights_only=True)
        return cache[name]
    return load
