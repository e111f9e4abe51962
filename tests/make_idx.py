"""Shared with oracle/make_golden.py: the deterministic sample positions used in feature fixtures."""
import numpy as np

SAMPLE_N = 64


def sample_idx(numel, salt):
    return np.random.RandomState(1000 + salt).randint(0, numel, size=SAMPLE_N)
