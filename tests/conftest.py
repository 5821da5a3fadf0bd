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
    # PyTorch first where a test also uses it (device tensors, streams): the C-ABI library then binds to the HIP runtime
    # torch already loaded; the other order leaves torch without a device ("No HIP GPUs are available")
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    import hnsw_rs_amd as H
    H.build_native()
    H.lib()
    return H


@pytest.fixture
def knob(monkeypatch):
    """knob(name, value) / knob(name, None): sets / removes an HNSWGPU_* hook in the environment and tells the library, which reads
    the environment once per process and otherwise only on hnswgpu_reload_env; both are undone when the test ends."""
    import hnsw_rs_amd as H

    def set_(name, value=None):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(value))
        H.reload_env()
    yield set_
    monkeypatch.undo()
    H.reload_env()


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


def probability(n, d, seed):
    """Probability vectors (non-negative, sum 1 in f32, some exact zeros): the inputs DistHellinger / DistJeffreys /
    DistJensenShannon are defined on."""
    x = np.random.default_rng(seed).random((n, d), dtype=np.float32) + np.float32(1e-3)
    x[:, ::5] = 0.0
    return np.ascontiguousarray((x / x.sum(1, dtype=np.float32)[:, None]).astype(np.float32))


def same_dump_after_reload(before_graph, after_graph, reloads=1):
    """A dump written by an index that was RELOADED from `before` equals `before` byte for byte except for the level
    scale (8 bytes at offset 6 of a v4 graph file): the reference reloads the dumped absolute scale as a factor of
    1/ln(M) (src/hnswio.rs:773-777, src/hnsw.rs:339-352) and dumps the product again (src/hnswio.rs:1365-1371)."""
    import math
    import struct
    a, b = open(before_graph, "rb").read(), open(after_graph, "rb").read()
    if len(a) != len(b) or a[:6] != b[:6] or a[14:] != b[14:]:
        return False
    m = a[5]
    sa, sb = struct.unpack("=d", a[6:14])[0], struct.unpack("=d", b[6:14])[0]
    want = sa
    for _ in range(reloads):
        want = want / math.log(float(max(2, m)))
    return sb == want
