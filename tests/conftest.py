import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'edm_golden.npz'))


@pytest.fixture(scope='session')
def manifest():
    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_full():
    """outputs of the reference itself at the full BASELINE network sizes (tests/golden/make_golden_fullsize.py)"""
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'fullsize_golden.npz'))


@pytest.fixture(scope='session')
def manifest_full():
    with open(os.path.join(ROOT, 'tests', 'golden', 'fullsize_manifest.json')) as f:
        return json.load(f)
