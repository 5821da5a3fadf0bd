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


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_plan_n_gpus_share_one_batch_of_100000():
    """BASELINE.json configs[3] as stated (src/hnsw.rs:1612-1635 answers ONE batch): N > 1 GPUs share the 100 000 queries, total work
    fixed = strong scaling; --weak / --nq keep a fixed count per GPU; N = 1 is config 2's 10 000."""
    b = _load_bench()
    cfg = b.CONFIGS["sift1m"]
    assert b.plan_queries(cfg, 1) == {"nq_local": 10000, "nq_total": 10000, "scaling": "weak", "mode": "one GPU, one batch", "dropped": 0}
    for n in (2, 4, 8):
        p = b.plan_queries(cfg, n)
        assert p["nq_total"] == 100000 and p["nq_local"] == 100000 // n and p["scaling"] == "strong" and p["dropped"] == 0
    p = b.plan_queries(cfg, 8, weak=True)
    assert p["nq_local"] == 12500 and p["nq_total"] == 100000 and p["scaling"] == "weak"
    p = b.plan_queries(cfg, 4, weak=True)
    assert p["nq_local"] == 12500 and p["nq_total"] == 50000 and p["scaling"] == "weak"
    p = b.plan_queries(cfg, 2, nq_override=500)
    assert p["nq_local"] == 500 and p["nq_total"] == 1000 and p["scaling"] == "weak"
    p = b.plan_queries(cfg, 3)  # a count 100 000 is not a multiple of: equal blocks, the remainder dropped and reported
    assert p["nq_local"] == 33333 and p["nq_total"] == 99999 and p["dropped"] == 1


@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_gpus_n_plan_through_the_launcher(n):
    """`python bench.py --gpus N --plan-only`: the command launches its N ranks (gloo, no device), every rank works the plan out and
    the blocks they would search are gathered: 100 000 queries in all, N contiguous blocks in input order, "strong"."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--plan-only", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == n and j["scaling"] == "strong"
    assert j["config"]["queries_total"] == 100000 and j["config"]["queries_per_gpu"] == 100000 // n
    assert j["blocks"] == [[i * (100000 // n), (i + 1) * (100000 // n)] for i in range(n)]


WORKER_OVERLAP = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import hnsw_rs_amd
    from hnsw_rs_amd.sharded import OverlappedExchange, shard_bounds
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nq_local, k = 37, 5
    nq = nq_local * world
    x = OverlappedExchange(nq, k, world, "cpu", "cpu")

    def fill(pk, step, r):          # what "the search" of rank r leaves for batch `step`
        base = 1000 * step + 100 * r
        pk.ids[:] = torch.arange(nq_local * k, dtype=torch.int64).view(nq_local, k) + base
        pk.dists[:] = float(base)
        pk.counts[:] = step + r

    def expect(step):
        ids = torch.cat([torch.arange(nq_local * k, dtype=torch.int64).view(nq_local, k) + 1000 * step + 100 * r for r in range(world)])
        counts = torch.cat([torch.full((nq_local,), step + r, dtype=torch.int32) for r in range(world)])
        return ids, counts

    ok = True
    # like bench.py: ranks first spin by the CLOCK without exchanging (different iteration counts per rank), then a common
    # number of steps with the exchange overlapped; a buffer is reused two steps later, after its exchange was waited for
    t0 = time.perf_counter(); spins = 0
    while time.perf_counter() - t0 < 0.05 * (1 + rank):
        fill(x.buffer(spins), 999, rank); spins += 1
    for step in range(9):
        pk = x.buffer(step)
        if step >= 2:                # the exchange of step - 2 used this pair: it is complete now, and its result intact
            ids, dd, counts = x.gatherers[step % 2].in_input_order()
            e_ids, e_counts = expect(step - 2)
            ok = ok and torch.equal(ids, e_ids) and torch.equal(counts, e_counts) and float(dd[0, 0]) == 1000.0 * (step - 2)
        fill(pk, step, rank)
        x.exchange(step, overlap=(step != 4))   # one of them not overlapped, as bench.py's gather_ms steps do
    ids, dd, counts = x.gathered(8)
    e_ids, e_counts = expect(8)
    ok = ok and torch.equal(ids, e_ids) and torch.equal(counts, e_counts)
    ids, _, _ = x.gathered(7)
    ok = ok and torch.equal(ids, expect(7)[0])
    t = torch.ones(1); dist.all_reduce(t)      # collectives still pair up afterwards
    ok = ok and int(t.item()) == world
    print("RANK", rank, "spins", spins, "OK" if ok else "MISMATCH", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)
""")


def test_overlapped_exchange_two_ranks_gloo(tmp_path):
    """hnsw_rs_amd.sharded.OverlappedExchange, the exchange bench.py --gpus N runs: two packed buffers alternate, the all-gather of
    batch i is asynchronous and overlaps batch i + 1, a buffer is rewritten only after its exchange completed; ranks that first
    spin by the clock WITHOUT exchanging (different counts per rank) stay paired afterwards -- the round-4 hang of the N = 2
    command was a clock-based warm-up that issued collectives."""
    script = tmp_path / "worker_overlap.py"
    script.write_text(WORKER_OVERLAP.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK") == 2
