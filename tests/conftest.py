"""pytest configuration: registers the `gpu` marker and puts the package dir on sys.path.

`-m "not gpu"` : oracle vs golden fixtures, host logic, C-ABI symbol checks, gloo tests (CPU).
`-m gpu`       : HIP path vs oracle / golden fixtures, through the C-ABI (needs an MI355X).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
PKG = os.path.join(ROOT, 'equi-articulated-pose_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope='session')
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _drain_gpu_between_tests(request):
    """GPU tests only: every test starts and ends on an idle device with the allocator's cache returned (the suite's largest tests
    hold 50 - 90 GB: they should not depend on what the tests before them left cached)."""
    if request.node.get_closest_marker('gpu') is None:
        yield
        return
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
    yield
    if torch.cuda.is_available():
        torch.cuda.synchronize()
