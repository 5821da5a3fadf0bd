"""N>1 path on CPU: world_size-2 gloo processes shard a query batch, search their shard (the oracle stands
in for the per-GPU searcher -- tests may use it), all-gather, and must reproduce the unsharded answers."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    import hnsw_rs_amd  # noqa: F401
    from hnsw_rs_amd.sharded import shard_bounds
    for nq in (0, 1, 7, 10, 12500, 100000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(nq, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == nq
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(100000, 8, 3) == (37500, 50000)  # BASELINE config 4: 8 x 12 500


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import hnsw_rs_amd
    from hnsw_rs_amd.sharded import sharded_parallel_search
    import oracle_lib
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    rng = np.random.default_rng(5)
    X = rng.random((1500, 12), dtype=np.float32)
    Q = rng.random((101, 12), dtype=np.float32)          # odd size: shards differ by one row
    o = oracle_lib.OracleHnsw(10, 1500, 16, 50, "DistL2"); o.insert_batch(X)   # replicated "index"
    def search_fn(q, k, ef):
        r = o.parallel_search(q, k, ef, 1)
        return (torch.from_numpy(r.ids.astype(np.int64)), torch.from_numpy(r.dists), torch.from_numpy(r.counts.astype(np.int32)))
    ids, dists, counts = sharded_parallel_search(search_fn, Q, 5, 20)
    ref = o.parallel_search(Q, 5, 20, 1)
    ok = (np.array_equal(ids.numpy().astype(np.uint64), ref.ids) and np.array_equal(dists.numpy(), ref.dists)
          and np.array_equal(counts.numpy().astype(np.uint32), ref.counts))
    print("RANK", rank, "OK" if ok else "MISMATCH", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)
""")


def test_two_rank_gloo_search_equals_unsharded(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK") == 2


def test_bench_gpus_n_is_a_plain_command():
    """`python bench.py --gpus N` (N > 1) launches its own ranks; with fewer HIP devices than N it ends at once with a
    clear message and a non-zero code -- never a hang, never a launcher requirement."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, r.stdout + r.stderr
    assert "HIP device" in r.stderr and not r.stdout.strip()
