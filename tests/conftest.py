import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'amt-saga_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def refvec():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'reference_vectors.npz'))


@pytest.fixture(scope='session', autouse=True)
def _blas_threads():
    """The oracle's convolutions are many small GEMMs: on a many-core host whose cgroup grants only
    a share of the cores, OpenBLAS' default (one thread per logical core) oversubscribes badly.
    Cap the pool for the tests (bench.py's cpu_baseline keeps the default and states its core count)."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        yield
        return
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    with threadpool_limits(limits=max(1, min(16, avail))):
        yield
