import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; oracle/dd_oracle.h)."""
    from oracle import dd_oracle
    dd_oracle.lib()
    return dd_oracle


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


_ENGINES = {}


@pytest.fixture(scope="session")
def engine_factory(torch_cuda):
    """Engines are cached per (log2m, canonical): contexts keep their HBM workspaces."""
    from dandd_amd.engine import Engine

    def make(log2m=14, canonical=True):
        key = (log2m, canonical)
        if key not in _ENGINES:
            _ENGINES[key] = Engine(device=0, log2m=log2m, canonical=canonical)
        return _ENGINES[key]

    yield make
    for e in _ENGINES.values():
        e.close()
    _ENGINES.clear()
