import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


os.environ.setdefault('DTS_GRAPHS_STRICT', '1')      # a refused HIP-graph capture fails a test instead of quietly measuring / checking eager launches


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


@pytest.fixture(scope='session')
def golden_c3():
    """the reference's own run of BASELINE configs[2] end to end (tests/golden/make_golden_config3.py, ~1.5 h of CPU)"""
    p = os.path.join(ROOT, 'tests', 'golden', 'config3_golden.npz')
    if not os.path.exists(p):
        pytest.skip('tests/golden/config3_golden.npz not generated')
    return np.load(p)


@pytest.fixture(scope='session')
def manifest_c3():
    p = os.path.join(ROOT, 'tests', 'golden', 'config3_manifest.json')
    if not os.path.exists(p):
        pytest.skip('tests/golden/config3_manifest.json not generated')
    with open(p) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_c3_seed0():
    """a second run of configs[2] by the reference, at a seed that was not scanned on the GPU first (make_golden_config3.py --seed 0 --prefix config3_seed0)"""
    p = os.path.join(ROOT, 'tests', 'golden', 'config3_seed0_golden.npz')
    if not os.path.exists(p):
        pytest.skip('tests/golden/config3_seed0_golden.npz not generated')
    return np.load(p)


@pytest.fixture(scope='session')
def manifest_c3_seed0():
    p = os.path.join(ROOT, 'tests', 'golden', 'config3_seed0_manifest.json')
    if not os.path.exists(p):
        pytest.skip('tests/golden/config3_seed0_manifest.json not generated')
    with open(p) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_mcts_full():
    """the reference's own MCTS search at full network size (tests/golden/make_golden_mcts_fullsize.py)"""
    p = os.path.join(ROOT, 'tests', 'golden', 'mcts_fullsize_golden.npz')
    if not os.path.exists(p):
        pytest.skip('tests/golden/mcts_fullsize_golden.npz not generated')
    with open(os.path.join(ROOT, 'tests', 'golden', 'mcts_fullsize_manifest.json')) as f:
        return np.load(p), json.load(f)
