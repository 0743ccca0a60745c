"""pytest configuration: `gpu` marker + shared fixture loaders."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: more than a minute of oracle (CPU) time; part of -m gpu")


@pytest.fixture(scope='session')
def golden_kernels():
    return np.load(os.path.join(GOLDEN, 'kernels.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_solves():
    return np.load(os.path.join(GOLDEN, 'solves.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_regression():
    return np.load(os.path.join(GOLDEN, 'regression_small.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_receivers():
    return np.load(os.path.join(GOLDEN, 'receivers.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_gridding():
    return np.load(os.path.join(GOLDEN, 'gridding.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_solves32():
    return np.load(os.path.join(GOLDEN, 'solves32.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_gradient():
    return np.load(os.path.join(GOLDEN, 'gradient.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_sources():
    """Source vectors of the reference's get_source_field incl. magnetic dipoles (tools/make_golden.py sources)."""
    return np.load(os.path.join(GOLDEN, 'sources.npz'), allow_pickle=False)
