import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (HIP device)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAVE_GPU = None


def pytest_collection_modifyitems(config, items):
    # no test may hang the run (multi-process / multi-thread tiling tests): with the
    # pytest-timeout plugin present, every test gets a ceiling
    if config.pluginmanager.hasplugin('timeout'):
        for item in items:
            if item.get_closest_marker('timeout') is None:
                item.add_marker(pytest.mark.timeout(900))
    # gpu-marked tests are selected with -m gpu on the GPU box; when collected
    # without a device (plain `pytest tests/`), skip them instead of failing.
    global HAVE_GPU
    gpu_items = [it for it in items if 'gpu' in it.keywords]
    if not gpu_items:
        return
    if HAVE_GPU is None:
        HAVE_GPU = _have_gpu()
    if not HAVE_GPU:
        skip = pytest.mark.skip(reason='no HIP device in this container')
        for it in gpu_items:
            it.add_marker(skip)
