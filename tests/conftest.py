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
