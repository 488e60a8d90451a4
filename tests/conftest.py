import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the wider fuzz slices (GPU, about two minutes together; part of -m gpu, deselect with -m 'gpu and not slow')")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py

    oracle_py.lib()
    return oracle_py


@pytest.fixture()
def params(oracle):
    p = oracle.default_params()
    p.scaling = 0  # plain ADMM for the twin comparisons; tests that exercise equilibration set it explicitly
    return p
