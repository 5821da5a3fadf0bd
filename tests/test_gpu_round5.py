"""Round-5 GPU tests: RCCL entered on the 1-GPU box (a one-rank process group running the N > 1 exchange).  The round's other GPU
tests live where they belong: the native peer gather of the sharded C-ABI call in tests/test_gpu_round4.py (config 4 at its own
size), the lane algorithms of the search kernels run on the device from scripts in tests/test_gpu_lane_lab.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, timeout=420):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_one_rank_rccl_group_runs_the_overlapped_exchange(tmp_path):
    """`bench.py --gpus 1 --exchange`: the process creates a ONE-rank `nccl` (= RCCL) process group and every step ends with the
    real OverlappedExchange -- asynchronous `all_gather_into_tensor` on the uint8 view of the packed answers, two buffers
    alternating -- exactly the code the 8-GPU run executes per rank.  The answers that come back through the collective equal
    the ones a run without any process group returns (src/hnsw.rs:1612-1635: answers in input order)."""
    cache = str(tmp_path / "cache")
    common = ["--gpus", "1", "--n", "20000", "--nq", "1000", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-recall",
              "--no-concurrent", "--no-boundary", "--no-traffic", "--cache-dir", cache]
    a1, a2 = str(tmp_path / "plain.npz"), str(tmp_path / "rccl.npz")
    r1, j1 = _bench(["--dump-answers", a1] + common)
    assert r1.returncode == 0 and j1 is not None, r1.stdout[-2000:] + r1.stderr[-4000:]
    assert j1["rccl"] is None and j1["config"]["exchange"] == "none"
    r2, j2 = _bench(["--exchange", "--backend", "nccl", "--dump-answers", a2] + common)
    assert r2.returncode == 0 and j2 is not None, r2.stdout[-2000:] + r2.stderr[-4000:]
    assert j2["n_gpus"] == 1
    assert j2["rccl"]["backend"] == "nccl" and j2["rccl"]["requested"] == "nccl" and j2["rccl"]["ranks_seen"] == 1
    assert j2["rccl"]["fallback_reason"] is None
    assert "RCCL" in j2["config"]["exchange"] and j2["gather_ms"] is not None and j2["gather_ms"] > 0
    print("one-rank RCCL exchange:", j2["rccl"], "gather_ms", j2["gather_ms"])
    one, two = np.load(a1), np.load(a2)
    assert np.array_equal(one["counts"], two["counts"])
    assert np.array_equal(one["ids"], two["ids"])
    assert np.array_equal(one["dists"].view(np.uint32), two["dists"].view(np.uint32))
