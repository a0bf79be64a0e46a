import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth_wave(seed, n):
    """Same generator as oracle/gen_golden.py:synth_wave."""
    return (np.random.RandomState(seed).randn(n) * 0.1).astype(np.float32)


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
