import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def native():
    """The product package with its C-ABI library built (cross-compiles without a GPU)."""
    import hnsw_rs_amd as H
    H.build_native()
    H.lib()
    return H


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


def uniform(n, d, seed):
    return np.random.default_rng(seed).random((n, d), dtype=np.float32)


def normalized(n, d, seed):
    x = uniform(n, d, seed)
    import oracle_lib
    for i in range(n):  # the crate's l2_normalize, f32
        oracle_lib.lib().orc_l2_normalize(x[i].ctypes.data, d)
    return x
