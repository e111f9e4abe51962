import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'multirank: spawns bench.py ranks as subprocesses (control flow of the N > 1 path; part of -m gpu, '
                                       'deselect with -m "gpu and not multirank" when gating a kernel change)')


def load_golden(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.fixture(scope='session')
def kat():
    return load_golden('kat_small.npz')
