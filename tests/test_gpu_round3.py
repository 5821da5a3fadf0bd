"""Round-3 GPU tests: the multi-rank bench command as the driver types it, full-size parity of BASELINE configs 2 and 3,
the row-resident norm of DistCosine, and the timed drop-in symbols."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, timeout=1200):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_2_as_a_plain_command_gathers_the_one_gpu_answers(tmp_path):
    """`python bench.py --gpus 2` with NO launcher: the command spawns its two ranks (here both on the box's one device),
    asks for RCCL, and the gathered answers of batch 0 equal what one GPU answers for the same 1000 queries.  RCCL refuses
    two ranks on one device; only then does the gather run over gloo, and the line says so (src/hnsw.rs:1612-1635)."""
    cache = str(tmp_path / "cache")
    common = ["--n", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-recall", "--no-concurrent", "--cache-dir", cache]
    a1, a2 = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    r1, j1 = _bench(["--gpus", "1", "--nq", "1000", "--dump-answers", a1] + common)
    assert r1.returncode == 0 and j1 is not None, r1.stdout[-2000:] + r1.stderr[-4000:]
    assert j1["n_gpus"] == 1 and j1["rccl"] is None and j1["config"]["exchange"] == "none"
    r2, j2 = _bench(["--gpus", "2", "--share-device", "--backend", "nccl", "--nq", "500", "--dump-answers", a2] + common)
    assert r2.returncode == 0 and j2 is not None, r2.stdout[-2000:] + r2.stderr[-4000:]
    assert j2["n_gpus"] == 2 and j2["config"]["queries_total"] == 1000 and j2["config"]["queries_per_gpu"] == 500
    assert j2["rccl"]["ranks_seen"] == 2 and j2["rccl"]["requested"] == "nccl"
    assert j2["rccl"]["backend"] in ("nccl", "gloo")
    if j2["rccl"]["backend"] == "gloo":   # allowed only as the answer to RCCL's refusal, which is then quoted
        assert j2["rccl"]["fallback_reason"]
    print("multi-rank gather ran over:", j2["rccl"])
    one, two = np.load(a1), np.load(a2)
    assert np.array_equal(one["counts"], two["counts"])
    assert np.array_equal(one["ids"], two["ids"])
    assert np.array_equal(one["dists"].view(np.uint32), two["dists"].view(np.uint32))
