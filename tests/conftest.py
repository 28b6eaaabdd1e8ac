import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock budgets (only run when selected with -m perf; never part of -m gpu)")


def pytest_collection_modifyitems(config, items):
    """wall-clock assertions do not belong in the correctness suite: a test marked `perf` runs only when the marker
    expression names it (`-m perf`), whatever other marks it carries"""
    if "perf" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="wall-clock budget: run with -m perf (throughput is reported by bench.py)")
    for item in items:
        if "perf" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sv_oracle

    sv_oracle.lib()
    return sv_oracle


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(scope="session")
def gpu():
    """cuda:0 device; fails (not skips) when a gpu-marked test runs without the HIP library or a GPU."""
    import torch

    import mrcc_amd

    mrcc_amd._lib.load()
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    return torch.device("cuda:0")
