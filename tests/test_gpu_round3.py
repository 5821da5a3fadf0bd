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


def _bench(args, timeout=420):
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
    # without --nq the N > 1 command is BASELINE config 4's plan: ONE batch shared by the ranks (8 000 queries for this small
    # shape), "strong", with the same batch on one GPU measured in the same invocation
    r3, j3 = _bench(["--gpus", "2", "--share-device", "--backend", "nccl", "--config", "random10k", "--steps", "2", "--warmup", "1",
                     "--no-cpu-baseline", "--no-recall", "--no-concurrent", "--cache-dir", cache])
    assert r3.returncode == 0 and j3 is not None, r3.stdout[-2000:] + r3.stderr[-4000:]
    assert j3["scaling"] == "strong" and j3["config"]["queries_total"] == 8000 and j3["config"]["queries_per_gpu"] == 4000
    assert j3["one_gpu_same_batch_queries_per_s"] > 0 and j3["speedup_over_one_gpu_same_batch"] > 0


# ------------------------------------------------------------------------------------------------- full size
from test_gpu_parity import assert_same  # noqa: E402
from test_gpu_round2 import _clustered  # noqa: E402


def _full_size(native, oracle, tmp_path, knob, n, d, m, efc, dist, k, ef, nq, n_dup, n_small_table, normalize=False):
    """BASELINE config at its REAL size: GPU-assisted build (the product's) -> hnswio dump -> product reload + upload and
    oracle reload of the same files -> the same queries through both: ids, f32 distance bits, p_ids, counts identical --
    strict as shipped, then with every pop taken from the literal candidate heap, then with a visited table far too small."""
    X = _clustered(n, d, 0x5EED0001)
    if normalize:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    if n_dup:  # exact duplicates (distinct ids): queries near them meet equal f32 distances for certain
        rng = np.random.default_rng(3)
        X[rng.choice(n, n_dup, replace=False)] = X[rng.choice(n, n_dup, replace=False)]
    hb = native.Hnsw(m, n, 16, efc, dist)
    hb.set_build_options(nthreads=0, gpu_device=0, gpu_window=0)
    hb.parallel_insert(X)
    assert hb.get_nb_point() == n
    hb.file_dump(tmp_path, "full")
    del hb
    h = native.HnswIo(tmp_path, "full").load_hnsw(dist)
    h.upload(0)
    o = oracle.OracleHnsw.load(tmp_path, "full", dist)
    Q = _clustered(nq, d, 0x5EED0002)
    if normalize:
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q[:200] = X[np.random.default_rng(4).choice(n, 200, replace=False)]  # 200 queries ARE points
    ref = o.parallel_search(Q, k, ef)
    res = h.parallel_search_flat(Q, k, ef)
    assert_same(res, ref)
    ties = h.last_tie_count()
    knob("HNSWGPU_EXACT_FIRST", "1")
    assert_same(h.parallel_search_flat(Q, k, ef), ref)
    knob("HNSWGPU_EXACT_FIRST", None)
    knob("HNSWGPU_HASH_BITS", "8")
    s = n_small_table
    assert_same(h.parallel_search_flat(Q[:s], k, ef), oracle.SearchResult(ref.ids[:s], ref.dists[:s], ref.layers[:s], ref.ranks[:s], ref.counts[:s]))
    knob("HNSWGPU_HASH_BITS", None)
    return ties


def test_full_size_parity_config2_1m_x_128(native, oracle, tmp_path, knob):
    """BASELINE config 2 itself: 1M x 128 L2, M=16, ef_c=200, ef=64, 10 000 clustered queries (20 id bits, 125-KB bitmap slices)."""
    ties = _full_size(native, oracle, tmp_path, knob, 1_000_000, 128, 16, 200, "DistL2", 10, 64, 10_000, 2000, 1500)
    assert ties > 0


def test_full_size_parity_config3_cosine_1m2_x_25(native, oracle, tmp_path, knob):
    """BASELINE config 3 itself: 1.2M x 25 DistCosine, M=24, ef=128 (two result slots per lane; 21 id bits; every point's f64
    norm inside its own 128-byte row)."""
    _full_size(native, oracle, tmp_path, knob, 1_200_000, 25, 24, 400, "DistCosine", 10, 128, 4000, 2000, 1000)


def test_full_size_parity_config3_dot_1m2_x_25(native, oracle, tmp_path, knob):
    """Config 3 the way the reference runs it: DistDot on L2-normalised vectors (examples/ann-glove25-angular.rs:81-82, :107-108)."""
    _full_size(native, oracle, tmp_path, knob, 1_200_000, 25, 24, 400, "DistDot", 10, 128, 4000, 2000, 1000, normalize=True)


# ------------------------------------------------------------------------------------------------- DistCosine: norm inside the row
@pytest.mark.parametrize("d", [1, 2, 25, 29, 30, 31, 32, 33, 62, 63, 64, 94, 100, 126, 127, 128, 130, 254])
def test_cosine_norm_in_the_row_or_beside_it(native, oracle, tmp_path, d):
    """A DistCosine row keeps its f64 squared norm in the last 8 bytes of its 128-byte-padded row when the padding has two
    free floats (d % 32 in 1..30), else in the side array: both layouts, searched and evaluated, equal the oracle bit for bit.
    Values include huge / tiny norms whose f64 bit patterns read as f32 are NaN / Inf / denormal in the padding."""
    rng = np.random.default_rng(100 + d)
    n = 3000
    X = rng.random((n, d), dtype=np.float32) + np.float32(0.05)
    if d > 1:  # (one dimension: every angle is 0, and rounding trips the crate's own assert on 1 - dot/norms >= -2e-5)
        X[::7] *= np.float32(1e18)      # norms whose high words look like odd floats
        X[1::7] *= np.float32(1e-18)
        X[5] = 0.0                      # a zero vector: DistCosine's 0-norm rule
    Q = rng.random((300, d), dtype=np.float32)
    Q[:50] = X[:50]
    o = oracle.OracleHnsw(12, n, 16, 80, "DistCosine")
    o.insert_batch(X)
    o.file_dump(tmp_path, "cos")
    h = native.HnswIo(tmp_path, "cos").load_hnsw("DistCosine")
    h.upload(0)
    assert_same(h.parallel_search_flat(Q, 10, 100), o.parallel_search(Q, 10, 100))
    got = native.eval_distance_matrix("DistCosine", Q[:8], X[:200], batch=33)
    want = oracle.dist_matrix("DistCosine", Q[:8], X[:200])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ------------------------------------------------------------------------------------------------- sharded, device-resident
def test_sharded_device_entry_point_equals_unsharded(native, oracle, tmp_path):
    """hnswgpu_search_batch_sharded_device: every shard's queries and answers already sit in HBM (here three shards on the
    box's one device, each with its own stream): the per-shard answers, put side by side, equal the unsharded search and the
    oracle (BASELINE config 4's partitioning)."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(12)
    n, d, k, ef = 6000, 24, 10, 48
    X = rng.random((n, d), dtype=np.float32)
    Q = rng.random((1001, d), dtype=np.float32)
    o = oracle.OracleHnsw(12, n, 16, 80, "DistL2")
    o.insert_batch(X)
    o.file_dump(tmp_path, "sh")
    h = native.HnswIo(tmp_path, "sh").load_hnsw("DistL2")
    lib = native.lib()
    dev = torch.device("cuda", 0)
    sizes = [334, 334, 333]
    bounds = np.cumsum([0] + sizes)
    qs = [torch.from_numpy(Q[bounds[s]:bounds[s + 1]]).to(dev) for s in range(3)]
    ids = [torch.zeros((sizes[s], k), dtype=torch.int64, device=dev) for s in range(3)]
    dists = [torch.zeros((sizes[s], k), dtype=torch.float32, device=dev) for s in range(3)]
    layers = [torch.zeros((sizes[s], k), dtype=torch.uint8, device=dev) for s in range(3)]
    ranks = [torch.zeros((sizes[s], k), dtype=torch.int32, device=dev) for s in range(3)]
    counts = [torch.zeros((sizes[s],), dtype=torch.int32, device=dev) for s in range(3)]
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    torch.cuda.synchronize(dev)

    def ptrs(ts):
        return (C.c_void_p * 3)(*[t.data_ptr() for t in ts])

    devices = (C.c_int * 3)(0, 0, 0)
    nqs = (C.c_uint64 * 3)(*sizes)
    rc = lib.hnswgpu_search_batch_sharded_device(h.handle, devices, 3, ptrs(qs), nqs, d, k, ef, ptrs(ids), ptrs(dists), ptrs(layers),
                                                 ptrs(ranks), ptrs(counts), (C.c_void_p * 3)(*[s.cuda_stream for s in streams]))
    assert rc == 0, native._native.last_error()
    torch.cuda.synchronize(dev)
    got = oracle.SearchResult(torch.cat(ids).cpu().numpy().astype(np.uint64), torch.cat(dists).cpu().numpy(), torch.cat(layers).cpu().numpy(),
                              torch.cat(ranks).cpu().numpy(), torch.cat(counts).cpu().numpy().astype(np.uint32))
    ref = o.parallel_search(Q, k, ef)
    assert_same(got, ref)
    assert_same(h.parallel_search_flat(Q, k, ef), ref)
    # the caller's current HIP device is what it was (every entry point restores it)
    assert torch.cuda.current_device() == 0
    # no layer / rank arrays, default streams
    rc = lib.hnswgpu_search_batch_sharded_device(h.handle, devices, 3, ptrs(qs), nqs, d, k, ef, ptrs(ids), ptrs(dists), None, None, ptrs(counts), None)
    assert rc == 0, native._native.last_error()
    torch.cuda.synchronize(dev)
    assert np.array_equal(torch.cat(ids).cpu().numpy().astype(np.uint64), ref.ids)


# ------------------------------------------------------------------------------------------------- construction: select on the device
@pytest.mark.parametrize("dist,d,m,efc,extend,keep", [
    ("DistL2", 16, 12, 60, False, False), ("DistL2", 8, 6, 40, False, True), ("DistL2", 8, 6, 40, True, False),
    ("DistCosine", 25, 8, 100, False, False), ("DistCosine", 33, 8, 60, False, True), ("DistL1", 10, 10, 40, False, False),
    ("DistDot", 12, 6, 250, False, False), ("DistJeffreys", 12, 8, 40, False, False), ("DistJensenShannon", 9, 10, 60, False, True)])
def test_window_1_construction_with_device_side_select_equals_the_serial_insertion(native, oracle, tmp_path, dist, d, m, efc, extend, keep):
    """GPU-assisted construction, one point per window, select_neighbours ON THE DEVICE (hnsw_build_select_kernel; on the host
    only when extend_candidates asks for the lists of the graph under construction): the dump must be the oracle's serially
    built graph byte for byte -- every search_layer AND every select_neighbours (src/hnsw.rs:1299-1421, keep_pruned included;
    dist(e, selected) with e as the first argument, which matters for the asymmetric Jeffreys / Jensen-Shannon sums)
    took the reference's decisions."""
    from conftest import normalized, probability, uniform
    n = 1200
    X = (probability(n, d, 77) if dist in ("DistJeffreys", "DistJensenShannon") else normalized(n, d, 77) if dist == "DistDot" else uniform(n, d, 77))
    o = oracle.OracleHnsw(m, n, 16, efc, dist)
    o.set_extend_candidates(extend)
    o.set_keeping_pruned(keep)
    o.insert_batch(X)
    o.file_dump(tmp_path, "orc")
    h = native.Hnsw(m, n, 16, efc, dist)
    h.set_extend_candidates(extend)
    h.set_keeping_pruned(keep)
    h.set_build_options(nthreads=1, gpu_device=0, gpu_window=1)
    h.parallel_insert(X)
    h.file_dump(tmp_path, "gpu")
    for ext in (".hnsw.graph", ".hnsw.data"):
        assert open(tmp_path / ("orc" + ext), "rb").read() == open(tmp_path / ("gpu" + ext), "rb").read()
