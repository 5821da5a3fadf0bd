#!/usr/bin/env python3
"""bench.py -- batched HNSW search (hnsw_rs `parallel_search_neighbours`) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched by torch.distributed.run with one rank per GPU.  A *step* is one pass of the hot path
over one batch of synthetic queries: greedy layer descent + layer-0 ef-expansion for every query
of the batch, inputs already resident in HBM.  Rank 0 prints ONE JSON line.

Workload at N=1 (BASELINE.json configs[1]): SIFT1M-shape synthetic, 1M x 128 f32, DistL2, M=16,
ef_construction=200, ef=64, k=10, 10 000 queries.  For N>1 the workload is BASELINE.json configs[3]
as it is stated: ONE batch of 100 000 queries (what `parallel_search` answers in one call,
src/hnsw.rs:1612-1635) split into N contiguous blocks of 100 000 / N, the graph replicated on every
GPU, the answers all-gathered over RCCL: the total work is fixed => "strong" scaling, and every N>1
line carries `one_gpu_same_batch_queries_per_s` -- the same 100 000 queries in one call on rank 0's
GPU, same invocation -- so that the 1 -> N ratio is read off one line.  `--weak` keeps the fixed
12 500 queries per GPU of the earlier rounds (per-GPU work fixed => "weak").

What is untimed setup: synthetic data, graph construction on the host cores (product builder,
cached as an hnswio dump under --cache-dir), dump reload, HBM upload, exact ground truth (torch
GEMM on the GPU: harness, not hot path).  The CPU oracle is used ONLY in the cpu_baseline leg.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

CONFIGS = {
    # name: n, d, dist, M, ef_c, k, ef, nq (N=1), nq per GPU (N>1, --weak), queries of the ONE batch N>1 GPUs share (strong scaling)
    "sift1m": dict(n=1_000_000, d=128, dist="DistL2", M=16, efc=200, k=10, ef=64, nq=10_000, nq_multi=12_500, nq_total_multi=100_000,
                   label="SIFT1M-shape synthetic 1Mx128 f32 L2 M=16 ef=64"),
    "glove25": dict(n=1_200_000, d=25, dist="DistCosine", M=24, efc=400, k=10, ef=128, nq=10_000, nq_multi=10_000, nq_total_multi=80_000,
                    label="GloVe-25-shape synthetic 1.2Mx25 f32 cosine M=24 ef=128"),
    # the reference's own choice for this data set: DistDot on L2-normalised vectors (examples/ann-glove25-angular.rs:81-82, :107-108)
    "glove25_dot": dict(n=1_200_000, d=25, dist="DistDot", M=24, efc=400, k=10, ef=128, nq=10_000, nq_multi=10_000, nq_total_multi=80_000,
                        label="GloVe-25-shape synthetic 1.2Mx25 f32, L2-normalised, DistDot M=24 ef=128"),
    "mnist784": dict(n=60_000, d=784, dist="DistL2", M=32, efc=400, k=10, ef=200, nq=10_000, nq_multi=10_000, nq_total_multi=80_000,
                     label="MNIST-784-shape synthetic 60kx784 f32 L2 M=32 ef=200"),
    # BASELINE config 5's shape with four times the points: 753 MB of vectors, beyond the 256 MiB Infinity Cache -- the line whose
    # roofline fraction is an HBM number (config 5 itself, 188 MB, is served by the cache: its line says so)
    "mnist784_hbm": dict(n=240_000, d=784, dist="DistL2", M=32, efc=400, k=10, ef=200, nq=10_000, nq_multi=10_000, nq_total_multi=80_000,
                         label="MNIST-784-shape synthetic 240kx784 f32 L2 M=32 ef=200 (config 5's shape, 4 x the points: out of the Infinity Cache)"),
    "random10k": dict(n=10_000, d=25, dist="DistL2", M=15, efc=200, k=10, ef=24, nq=1_000, nq_multi=1_000, nq_total_multi=8_000,
                      label="random.rs shape 10kx25 f32 L2 M=15 ef=24"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def plan_queries(cfg, world, nq_override=0, weak=False):
    """Who searches what.  N = 1: the config's batch.  N > 1: BASELINE.json configs[3] as stated -- ONE batch (`nq_total_multi`
    queries: 100 000 for the SIFT1M shape) cut into `world` contiguous blocks of equal size, total work fixed: "strong" scaling
    (a total that `world` does not divide loses its last few queries, and says so).  --weak: `nq_multi` queries per GPU whatever
    N is (per-GPU work fixed).  --nq overrides the per-GPU count and makes the run a weak one."""
    if world == 1:
        nq_local = nq_override or cfg["nq"]
        return {"nq_local": nq_local, "nq_total": nq_local, "scaling": "weak", "mode": "one GPU, one batch", "dropped": 0}
    if nq_override or weak:
        nq_local = nq_override or cfg["nq_multi"]
        return {"nq_local": nq_local, "nq_total": nq_local * world, "scaling": "weak",
                "mode": f"{nq_local} queries per GPU whatever N is (--weak / --nq)", "dropped": 0}
    total = cfg["nq_total_multi"]
    nq_local = total // world
    return {"nq_local": nq_local, "nq_total": nq_local * world, "scaling": "strong",
            "mode": f"one batch of {total} queries in {world} contiguous blocks of {nq_local} (BASELINE.json configs[3])",
            "dropped": total - nq_local * world}


def synth(n, d, seed, kind):
    """Deterministic synthetic f32 vectors.  'uniform': iid U[0,1) (what every reference test uses);
    'clustered': 1000-centre Gaussian mixture, sigma=0.1 (BASELINE.md distribution B)."""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, d), dtype=np.float32)
    centres = np.random.default_rng(0xC0FFEE).random((1000, d), dtype=np.float32)
    out = np.empty((n, d), np.float32)
    step = 1 << 18
    for s in range(0, n, step):
        e = min(n, s + step)
        c = rng.integers(0, 1000, e - s)
        out[s:e] = centres[c] + 0.1 * rng.standard_normal((e - s, d), dtype=np.float32)
    return out


def spin_up(fn, seconds=0.3):
    """Untimed: calls fn until `seconds` have passed -- GPU clocks are back up after an idle period (see main)."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()


def ground_truth(torch, Xd, Qd, k, dist):
    """Exact k-NN on the GPU: GEMM shortlist of 4k candidates, then f64 re-evaluation of the metric."""
    nq = Qd.shape[0]
    short = min(4 * k, Xd.shape[0])
    ids = torch.empty((nq, k), dtype=torch.int64, device=Qd.device)
    dd = torch.empty((nq, k), dtype=torch.float64, device=Qd.device)
    xn = (Xd * Xd).sum(1)
    for s in range(0, nq, 2048):
        q = Qd[s:s + 2048]
        if dist == "DistL2":
            approx = xn[None, :] - 2.0 * (q @ Xd.T)
        else:  # cosine / dot: larger similarity = smaller distance
            approx = -(q @ Xd.T) / (xn.sqrt()[None, :] if dist == "DistCosine" else 1.0)
        cand = approx.topk(short, dim=1, largest=False).indices
        xc = Xd[cand].double()
        qq = q.double()[:, None, :]
        if dist == "DistL2":
            ex = ((xc - qq) ** 2).sum(-1).sqrt()
        elif dist == "DistCosine":
            ex = 1.0 - (xc * qq).sum(-1) / ((xc * xc).sum(-1) * (qq * qq).sum(-1)).sqrt()
        else:
            ex = 1.0 - (xc * qq).sum(-1)
        top = ex.topk(k, dim=1, largest=False)
        ids[s:s + 2048] = torch.gather(cand, 1, top.indices)
        dd[s:s + 2048] = top.values
    return ids, dd


def numa_nodes():
    """{node: cpulist} of the host, from sysfs (empty when it cannot be read)."""
    import glob
    out = {}
    for p in sorted(glob.glob("/sys/devices/system/node/node[0-9]*/cpulist")):
        try:
            out[p.split("/")[-2]] = open(p).read().strip()
        except OSError:
            pass
    return out


def cpu_budget():
    """(logical CPUs visible, CPUs' worth of time this process may use, where that number comes from).  A container shows every
    logical CPU of its host but may be held to a CFS quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us): threads beyond the quota
    are throttled, not run -- Rayon's default pool (std::thread::available_parallelism) honours the quota too."""
    logical = os.cpu_count() or 1
    usable, source = logical, "all logical CPUs"
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < usable:
            usable, source = aff, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(quota) > 0 and -(-int(quota) // int(period)) < usable:
            usable, source = -(-int(quota) // int(period)), f"cgroup v2 cpu.max = {quota} {period}"
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and -(-quota // period) < usable:
                usable, source = -(-quota // period), f"cgroup v1 cfs quota {quota} / {period}"
        except (OSError, ValueError):
            pass
    return logical, max(1, usable), source


def cpu_baseline_leg(cache_dir, base, dist, Q, k, ef, res_ids, res_dists, st, cnt, cpu_seconds, orc=None):
    """Times the oracle (CPU restatement of the reference, test infrastructure) on the same graph and queries and
    compares the device answers with it.  Returns (cpu_baseline, parity) for the bench line.
    Protocol: wall time of the whole batched call, 1 warm-up + median of 5, on a sample of the batch sized
    for about `cpu_seconds` of CPU work in total."""
    import oracle_lib
    nq_local = Q.shape[0]
    logical, cores, budget_source = cpu_budget()  # cores: what this process may actually use (the box of round 5 showed 256, granted 16)
    if orc is None:
        t0 = time.time()
        orc = oracle_lib.OracleHnsw.load(cache_dir, base, dist)
        log(f"oracle reloaded the same dump in {time.time() - t0:.1f} s")
    log(f"timing the oracle's parallel_search on {cores} threads")
    probe = min(nq_local, 256)
    r = orc.parallel_search(Q[:probe], k, ef, cores)
    rate = probe / max(r.elapsed_s, 1e-6)
    # Rayon's default is one thread per logical core; on a two-socket box the Arc refcounts of hub nodes bounce between the
    # sockets and the rate FALLS with the thread count, so 32 / 64 / 128 / 256 threads (and all logical CPUs) are timed, each
    # worker pinned to its own logical CPU, NUMA node by NUMA node (oracle/pinning.hpp: T threads then span as few memory
    # domains as T allows), plus the unpinned run on all CPUs and on a quarter of them; the best rate is reported.
    # (thread counts around the CPU budget: a quarter, half, all of it, and twice it -- what an oversubscribed pool costs)
    threads = sorted({max(1, cores // 4), max(1, cores // 2), cores, min(logical, 2 * cores)}, reverse=True)
    unpinned_threads = sorted({cores, max(1, cores // 4)}, reverse=True)
    # ~6 runs each of: the pinned thread counts, two SIMD-order ones, two unpinned ones: size the sample for the budget
    sample = int(min(nq_local, max(probe, rate * cpu_seconds / (6.0 * (len(threads) + 4)))))

    def median_of_5(fn):
        fn()  # warm-up
        return float(np.median([sample / fn().elapsed_s for _ in range(5)]))

    unpinned = {nt: median_of_5(lambda nt=nt: orc.parallel_search(Q[:sample], k, ef, nt)) for nt in unpinned_threads}
    oracle_lib.OracleHnsw.set_thread_pinning(True)
    trials = {nt: median_of_5(lambda nt=nt: orc.parallel_search(Q[:sample], k, ef, nt)) for nt in threads}
    oracle_lib.OracleHnsw.set_thread_pinning(False)
    r = orc.parallel_search(Q, k, ef, cores)  # scalar (reference-order) answers of the WHOLE batch for the parity check
    best_threads = max(trials, key=trials.get)
    cpu_qps = trials[best_threads]
    placement = "pinned"
    if max(unpinned.values()) > cpu_qps:
        best_threads = max(unpinned, key=unpinned.get)
        cpu_qps, placement = unpinned[best_threads], "unpinned"
    # the reference's published numbers use its SIMD feature build: the same search with the distances summed in
    # the crate's 8-lane order (timing only; last-bit differences, so parity below uses the scalar answers)
    simd_threads_tried = sorted(trials, key=trials.get, reverse=True)[:2]
    orc.set_simd_order(True)
    oracle_lib.OracleHnsw.set_thread_pinning(True)
    simd_trials = {nt: median_of_5(lambda nt=nt: orc.parallel_search(Q[:sample], k, ef, nt)) for nt in simd_threads_tried}
    oracle_lib.OracleHnsw.set_thread_pinning(False)
    orc.set_simd_order(False)
    simd_threads = max(simd_trials, key=simd_trials.get)
    arithmetic = "scalar (bit-exact order)"
    if simd_trials[simd_threads] > cpu_qps:
        cpu_qps, best_threads, placement = simd_trials[simd_threads], simd_threads, "pinned"
        arithmetic = "approximate simd-order (8 vertical f32 lanes summed left to right: an unpinned restatement of the crate's simdeez_f build)"
    # for honesty: the same search freed of the reference's data model (flat arrays, epoch visited array, SIMD-order
    # sums, prefetch: oracle/flat_baseline.hpp) -- what the host cores can do, NOT the reference's cost structure
    flat = orc.flat_baseline()
    flat_sample = int(min(nq_local, sample * 8))
    flat_trials = {}
    for nt in unpinned_threads:
        flat.parallel_search(Q[:flat_sample], k, ef, nt)
        flat_trials[nt] = float(np.median([flat_sample / flat.parallel_search(Q[:flat_sample], k, ef, nt).elapsed_s for _ in range(5)]))
    flat_threads = max(flat_trials, key=flat_trials.get)
    fr = flat.parallel_search(Q[:sample], k, ef, cores)
    flat_agree = float(np.mean(fr.ids == r.ids[:sample]))
    del flat
    # Parity at full size.  status 2 = the kernel met a decision that depends on the reference's heap order and did not
    # resolve it (strict ties off), 3 = resolved with the literal heaps (DESIGN.md "ties").
    gpu_ids = res_ids.astype(np.uint64)
    gpu_bits = np.ascontiguousarray(res_dists, dtype=np.float32).view(np.uint32)
    tie_flag = (st[:, 3] == 2) | (st[:, 3] == 3)
    exact_used = st[:, 3] == 3
    row_ids_ok = np.all(r.ids == gpu_ids, axis=1) & (r.counts == cnt.astype(np.uint32))
    row_bits_ok = np.all(r.dists.view(np.uint32) == gpu_bits, axis=1)
    parity = {"queries_checked": int(nq_local),
              "all_ids_identical": bool(row_ids_ok.all()),
              "all_f32_distance_bits_identical": bool(row_bits_ok.all()),
              "tie_free_queries": int((~tie_flag).sum()),
              "tie_free_ids_identical": bool(row_ids_ok[~tie_flag].all()),
              "tie_free_f32_distance_bits_identical": bool(row_bits_ok[~tie_flag].all()),
              "queries_that_met_equal_distances": int((st[:, 7] & 1).sum()),
              "queries_whose_answer_depends_on_heap_order": int(tie_flag.sum()),
              "resolved_with_literal_heaps": int(exact_used.sum()),
              "heap_order_queries_ids_identical": int(row_ids_ok[tie_flag].sum()),
              "heap_order_queries_distance_bits_identical": int(row_bits_ok[tie_flag].sum())}
    cpu_baseline = {"value": round(cpu_qps, 1), "unit": "queries/s", "cores": best_threads, "kind": "port",
                    "arithmetic": arithmetic, "protocol": "1 warm-up + median of 5 batched calls",
                    "by_threads": {str(t): round(v, 1) for t, v in trials.items()},
                    "by_threads_unpinned": {str(t): round(v, 1) for t, v in unpinned.items()},
                    "by_threads_simd_order": {str(t): round(v, 1) for t, v in simd_trials.items()},
                    "placement": placement,
                    "pinning": "by_threads / by_threads_simd_order: worker t pinned to the t-th logical CPU, NUMA node by NUMA node "
                               "(oracle/pinning.hpp; first CPUs: %s); by_threads_unpinned: the scheduler's placement" % [oracle_lib.OracleHnsw.pinning_cpu(t) for t in (0, 1, 2, 3)],
                    "numa_nodes": numa_nodes(), "logical_cpus": logical, "usable_cpus": cores, "usable_cpus_from": budget_source,
                    "index_pages": "interleaved over the NUMA nodes while one thread loads the index (oracle/pinning.hpp: set_mempolicy)",
                    "allocator": "glibc malloc (per-thread arenas) -- also what a Rust binary uses by default (std's System allocator)",
                    "flat_avx": {"value": round(flat_trials[flat_threads], 1), "cores": flat_threads,
                                 "by_threads": {str(t): round(v, 1) for t, v in flat_trials.items()},
                                 "ids_agreeing_with_the_port": round(flat_agree, 4),
                                 "note": "same algorithm on flat arrays with SIMD-order sums and prefetch (oracle/flat_baseline.hpp): "
                                         "an optimised CPU implementation, not the reference's data model"},
                    "sample": f"first {sample} of the same {nq_local} queries (flat variant: {flat_sample}), same graph (reloaded from the "
                              f"same hnswio dump), oracle parallel_search (Rayon-style worker threads; best of {threads} pinned / {unpinned_threads} unpinned threads on a "
                              f"{logical}-CPU host that grants this process {cores} CPUs ({budget_source}), scalar and SIMD-order distances)"}
    del orc
    return cpu_baseline, parity


def probe_rccl_on_one_device(n_ranks, seconds=90):
    """Does RCCL complete a collective with n_ranks ranks on ONE HIP device?  Asked in disposable processes, because its way
    of refusing may be a hang.  Returns (True, None) or (False, reason)."""
    import signal
    import socket
    import subprocess
    import tempfile
    code = ("import datetime, torch, torch.distributed as dist\n"
            "torch.cuda.set_device(0)\n"
            "dist.init_process_group('nccl', timeout=datetime.timedelta(seconds=45), device_id=torch.device('cuda', 0))\n"
            "t = torch.ones(1, device='cuda:0'); dist.all_reduce(t); torch.cuda.synchronize()\n"
            "assert int(t.item()) == dist.get_world_size()\n"
            "dist.destroy_process_group()\n")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "rccl_probe.py")
        with open(script, "w") as f:
            f.write(code)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr",
                              "127.0.0.1", "--master-port", str(port), script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             text=True, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=seconds)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)  # the probe's own process group, nothing else
            p.communicate()
            return False, f"RCCL did not complete an all_reduce of {n_ranks} ranks sharing HIP device 0 within {seconds} s (it hangs instead of refusing)"
        if p.returncode == 0:
            return True, None
        text = [l.strip() for l in (out or "").splitlines() if l.strip() and "amdgpu.ids" not in l]
        for l in text[-25:]:
            log("rccl probe | " + l[:300])
        keys = ("Duplicate GPU", "ncclInvalidUsage", "invalid usage", "NCCL error", "ncclSystemError", "ncclUnhandled", "DistBackendError",
                "RuntimeError", "HIP error", "Error")
        lines = []
        for key in keys:  # the most telling line there is, never one of torch's shutdown warnings
            lines = [l for l in text if key in l and "Warning" not in l and "WARNING" not in l and "traceback" not in l]
            if lines:
                break
        return False, ("RCCL refused %d ranks on one device: %s" % (n_ranks, (lines[-1] if lines else "exit code %d" % p.returncode)[:240]))


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: this process becomes the launcher -- one rank per GPU through
    torch.distributed.run on this node -- and returns the ranks' exit code.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if args.plan_only:
        ndev = args.gpus  # (the plan needs no device: the ranks meet over gloo)
    if ndev == 0:
        print("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)", file=sys.stderr)
        return 2
    if ndev < args.gpus and not args.share_device:
        print(f"bench.py --gpus {args.gpus}: only {ndev} HIP device(s) visible on this node (one rank per GPU; "
              f"--share-device puts every rank on device 0 to exercise the multi-rank path on a smaller box)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    if args.share_device and args.backend == "nccl":
        # every rank on ONE device is not a configuration RCCL serves; whether it refuses (or hangs) is found out in throw-away
        # processes, and only then do the ranks gather over gloo -- the line reports which backend ran and why
        ok, reason = probe_rccl_on_one_device(args.gpus)
        if not ok:
            log(f"--share-device: {reason}; the answers are gathered over gloo")
            env["HNSW_BENCH_GATHER_FALLBACK"] = reason
    # this command's own arguments travel in the environment: the launcher's option parser would try to claim the ones that
    # look like prefixes of its own options (--n ...)
    env["HNSW_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)]
    log(f"--gpus {args.gpus}: launching {args.gpus} ranks ({' '.join(cmd[1:9])} ...)")
    return subprocess.call(cmd, env=env)


def filtered_measure(torch, H, lib, index, orc, n, d, k, ef, Q, cpu_queries, pcts, seed=0xF117):
    """Row f3 measured like the main row: Hnsw::search_filter with a sorted id vector allowing pct % of the points, the queries
    Q in one call with everything resident in HBM (hnswgpu_search_batch_filtered_device): queries/s, the kernel's time (HIP events),
    per-query work counters -> algorithmic bytes -> fraction of the HBM peak; and, when the oracle is given, its search_filter
    on the first cpu_queries of the same queries (all host threads): the row's CPU baseline, plus parity of those answers."""
    import ctypes as C
    dev = torch.device("cuda", torch.cuda.current_device())
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    nq = Q.shape[0]
    Qd = torch.from_numpy(Q).to(dev)
    ids = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    dists = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    layer = torch.zeros((nq, k), dtype=torch.uint8, device=dev)
    rank = torch.zeros((nq, k), dtype=torch.int32, device=dev)
    counts = torch.zeros((nq,), dtype=torch.int32, device=dev)
    stats = torch.zeros((nq, 8), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    rng = np.random.default_rng(seed)
    out = {}
    for pct in pcts:
        allowed = np.sort(rng.choice(n, max(1, n * pct // 100), replace=False)).astype(np.uint64)  # origin ids = 0..n-1 here
        ad = torch.from_numpy(allowed.view(np.int64)).to(dev)
        panics = C.c_uint32(0)

        def call():
            rc = lib.hnswgpu_search_batch_filtered_device(index.handle, Qd.data_ptr(), nq, d, k, ef, ad.data_ptr(), len(allowed), ids.data_ptr(),
                                                          dists.data_ptr(), layer.data_ptr(), rank.data_ptr(), counts.data_ptr(),
                                                          stats.data_ptr(), stream.cuda_stream, C.byref(panics))
            if rc != 0:
                raise RuntimeError(H._native.last_error())
        spin_up(call, 0.2)
        ts, kms = [], []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            call()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
            kms.append(index.last_search_kernel_ms())
        st = stats.cpu().numpy().astype(np.int64)
        if os.environ.get("HNSW_BENCH_DUMP_FILTERED"):  # profiling builds of the literal kernel put phase ticks into the counters
            np.save(os.path.join(os.environ["HNSW_BENCH_DUMP_FILTERED"], f"filtered_{nq}_{pct}pct.npy"), st)
        nd, nx, ni = st[:, 0], st[:, 1], st[:, 2]
        dur_us = ((st[:, 5] - st[:, 4]) & 0xFFFFFFFF) / 100.0  # 10 ns ticks
        alg = int(nd.sum()) * d * 4 + int(ni.sum()) * 4 + int(nx.sum()) * 8 + nq * (d * 4 + k * 12)
        k_ms = float(np.median(kms))
        e = {"allowed_points": int(len(allowed)), "queries_per_call": nq, "queries_per_s": round(nq / float(np.median(ts)), 1),
             "kernel": "hnsw_search_exact_kernel", "kernel_ms": round(k_ms, 3),
             "per_query": {"n_expand_p50_p99_max": [int(np.percentile(nx, 50)), int(np.percentile(nx, 99)), int(nx.max())],
                           "n_dist_mean": float(nd.mean()), "n_ids_read_mean": float(ni.mean()),
                           "candidate_heap_entries_left_p50_max": [int(np.percentile(st[:, 6], 50)), int(st[:, 6].max())],
                           "duration_us_p50_p99_max": [round(float(np.percentile(dur_us, 50)), 1), round(float(np.percentile(dur_us, 99)), 1), round(float(dur_us.max()), 1)],
                           "us_per_expansion_p50": round(float(np.median(dur_us / np.maximum(nx, 1))), 2)},
             "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": round(alg / (k_ms * 1e-3) / 1e9, 1),
             "frac_of_hbm_peak": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "reference_panics": int(panics.value)}
        if orc is not None and cpu_queries > 0:
            m = min(nq, cpu_queries)
            _, cores, _ = cpu_budget()  # (the CPUs this process is granted, not the ones it can see)
            best = None
            for nt in sorted({cores, min(os.cpu_count() or 1, 2 * cores)}):
                r = orc.parallel_search_filter(Q[:m], k, ef, allowed, nt)
                if best is None or r.elapsed_s < best[0]:
                    best = (r.elapsed_s, nt, r)
            el, nt, r = best
            g_ids, g_d, g_c = ids.cpu().numpy().astype(np.uint64)[:m], dists.cpu().numpy()[:m], counts.cpu().numpy().astype(np.uint32)[:m]
            same = bool(np.array_equal(g_c, r.counts) and all(np.array_equal(g_ids[i, :g_c[i]], r.ids[i, :g_c[i]]) and
                        np.array_equal(g_d[i, :g_c[i]].view(np.uint32), r.dists[i, :g_c[i]].view(np.uint32)) for i in range(m)))
            e["cpu_baseline"] = {"value": round(m / el, 1), "unit": "queries/s", "cores": nt, "kind": "port",
                                 "sample": f"first {m} of the same queries, oracle search_filter on worker threads (one call)"}
            e["parity_vs_oracle"] = {"queries_checked": m, "ids_distance_bits_counts_identical": same}
        out[f"{pct}pct"] = e
    return out


def live_traffic(args, timeout_s=120):
    """roofline.traffic measured in THIS invocation, on this box: PMC counters cannot be collected inside a running process, so a
    short child run of this script (the cached index, strict launches only, nothing else) is put under `rocprofv3 --pmc`, in two
    passes -- FETCH_SIZE (+ TCC_EA0_RDREQ_sum as a cross-check) and WRITE_SIZE do not fit one pass, and counters are never mixed
    with trace domains -- and the counters are averaged per dispatch of the strict search kernel in the default arithmetic.
    Corrected as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE is in KB and, on gfx950, counts 128-byte requests at
    64 bytes: bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 (fabric side: Infinity-Cache hits included).  Returns
    (bytes per launch or None, details)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return None, {"skipped": "rocprofv3 not found"}
    if any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, {"skipped": "this process is itself being profiled (rocprofv3 environment found): no nested counter runs"}
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", args.config, "--data", args.data, "--steps", "3", "--warmup", "1",
             "--cache-dir", args.cache_dir, "--batches", str(args.batches), "--no-cpu-baseline", "--no-recall", "--no-concurrent",
             "--no-boundary", "--no-traffic"]
    if args.n:
        child += ["--n", str(args.n)]
    if args.nq:
        child += ["--nq", str(args.nq)]
    if args.ef:
        child += ["--ef", str(args.ef)]
    out = tempfile.mkdtemp(prefix="hnsw_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("HNSW_BENCH_ARGV", None)
    counters = {}
    try:
        for name, ctrs in (("fetch", ["FETCH_SIZE", "TCC_EA0_RDREQ_sum"]), ("write", ["WRITE_SIZE"])):
            r = subprocess.run([tool, "--pmc", *ctrs, "-d", os.path.join(out, name), "--output-format", "csv", "--", *child],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None, {"skipped": f"rocprofv3 --pmc {' '.join(ctrs)} exited with {r.returncode}: {(r.stderr or '')[-300:]}"}
            per = {}
            for f in glob.glob(os.path.join(out, name, "**", "*_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    kn = row["Kernel_Name"]
                    i = kn.find("hnsw_search_kernel<")
                    if i < 0:
                        continue
                    targs = kn[i + len("hnsw_search_kernel<"):].split(">")[0].split(",")
                    if targs[-1].strip() != "true" or int(targs[0]) >= 7:   # strict launches in the default (scalar-order) arithmetic
                        continue
                    key = (f, row["Dispatch_Id"], row["Counter_Name"])
                    per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
            for (_f, _d, c), v in per.items():
                counters.setdefault(c, []).append(v)
    except subprocess.TimeoutExpired:
        return None, {"skipped": f"a rocprofv3 --pmc pass did not finish in {timeout_s} s"}
    except Exception as e:  # noqa: BLE001 -- an informational figure: never at the expense of the line
        return None, {"skipped": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(out, ignore_errors=True)
    if not counters.get("FETCH_SIZE") or not counters.get("WRITE_SIZE"):
        return None, {"skipped": "no dispatch of the strict search kernel in the counter output"}
    avg = {c: float(np.mean(v)) for c, v in counters.items()}
    nbytes = int(avg["FETCH_SIZE"] * 1024 * 2 + avg["WRITE_SIZE"] * 1024)
    return nbytes, {"FETCH_SIZE_KB": round(avg["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(avg["WRITE_SIZE"], 1),
                    "TCC_EA0_RDREQ": round(avg.get("TCC_EA0_RDREQ_sum", float("nan")), 1),
                    "fetch_bytes_from_rdreq_x128": int(avg["TCC_EA0_RDREQ_sum"] * 128) if "TCC_EA0_RDREQ_sum" in avg else None,
                    "dispatches_averaged": {c: len(v) for c, v in counters.items()},
                    "how": "this invocation, this box: two `rocprofv3 --pmc` passes (FETCH_SIZE TCC_EA0_RDREQ_sum | WRITE_SIZE) over a child run "
                           "`bench.py --steps 3 --warmup 1` (same config, cached index, all 4 query batches), per dispatch of the strict "
                           "search kernel; bytes = FETCH_SIZE (KB) x 1024 x 2 (gfx950: 128-byte requests tallied at 64) + WRITE_SIZE (KB) x 1024; "
                           "fabric-side requests, Infinity-Cache hits included"}


def boundary_timings(H, lib, index, cache_dir, base, dist_name, Q, k, ef, n, reps=25):
    """What a caller pays above the device-buffer call (reported next to `value`, never as `value`):
    * host_buffers: hnswgpu_search_batch -- pageable host matrices in, host arrays out (the queries read across PCIe by the descent
      kernel, the answers written into a pinned arena and unpacked), output arrays reused / allocated per call;
    * ffi: the reference's own symbol parallel_search_neighbours_f32 (src/libext.rs:205-254) on a handle loaded the
      reference's way (get_hnswio + load_hnswdump_f32_<Dist>): array of row pointers in, Vec_api<Neighbourhood_api> out --
      written in place by the search kernels (a page-locked slab) --, freed with hnswgpu_free_neighbourhood_vec;
    * filtered: Hnsw::search_filter with a sorted id vector allowing 1 % / 30 % of the points (literal-heap kernel)."""
    nq, d = Q.shape
    out = {}

    def rate(fn, nrep=reps):
        fn()         # (the first call of an entry point creates its workspace: hundreds of ms that are not the spin-up's)
        spin_up(fn)  # (this block follows host-only work: see spin_up)
        ts = []
        for _ in range(nrep):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return nq / float(np.median(ts))

    reuse = index.parallel_search_flat(Q, k, ef)  # a caller in steady state writes the same output arrays again
    out["host_buffers_queries_per_s"] = round(rate(lambda: index.parallel_search_flat(Q, k, ef, out=reuse)), 1)
    out["host_buffers_fresh_output_arrays_queries_per_s"] = round(rate(lambda: index.parallel_search_flat(Q, k, ef)), 1)
    loader = getattr(lib, "load_hnswdump_f32_" + dist_name, None)
    if loader is not None:
        cwd = os.getcwd()
        os.chdir(cache_dir)  # get_hnswio names a dump in the current directory (src/libext.rs:28-33)
        try:
            api = loader(lib.get_hnswio(len(base), base.encode()))
        finally:
            os.chdir(cwd)
        if api:
            rows = (C.c_void_p * nq)(*[Q.ctypes.data + i * d * 4 for i in range(nq)])
            first = {}

            def ffi_call():
                v = lib.parallel_search_neighbours_f32(api, nq, d, rows, k, ef)
                if not v:
                    raise RuntimeError("parallel_search_neighbours_f32 returned NULL: " + H._native.last_error())
                if not first:
                    first["ids0"] = [v.contents.ptr[0].neighbours[j].id for j in range(v.contents.ptr[0].nbgh)]
                lib.hnswgpu_free_neighbourhood_vec(v)

            out["ffi_parallel_search_neighbours_f32_queries_per_s"] = round(rate(ffi_call), 1)
            out["ffi_first_answer_ids"] = first.get("ids0")
            # Hnsw::search, one query per call (search_neighbours_f32, src/libext.rs:728-767; examples/random.rs:66-79 is exactly
            # 100 such calls): host pointer in, Neighbourhood_api out -- one wavefront on the device, descent + search kernel
            lat = []
            for i in range(120):
                qp = Q.ctypes.data + (i % nq) * d * 4
                t0 = time.perf_counter()
                one = lib.search_neighbours_f32(api, d, qp, k, ef)
                lat.append(time.perf_counter() - t0)
                if not one:
                    raise RuntimeError("search_neighbours_f32 returned NULL: " + H._native.last_error())
                lib.hnswgpu_free_neighbourhood(one)
            lat = np.array(lat[20:]) * 1e6  # (the first calls create the one-query workspace)
            out["single_query_latency_us"] = {"p50": round(float(np.percentile(lat, 50)), 1), "p99": round(float(np.percentile(lat, 99)), 1),
                                              "calls": int(len(lat)), "entry_point": "search_neighbours_f32 (host pointer in, Neighbourhood_api out)"}
            lib.drop_hnsw_f32(api)
    return out



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="sift1m", choices=sorted(CONFIGS))
    ap.add_argument("--data", default="clustered", choices=["clustered", "uniform"])
    ap.add_argument("--n", type=int, default=0, help="override the number of points (invalidates the headline config)")
    ap.add_argument("--nq", type=int, default=0, help="override queries per GPU")
    ap.add_argument("--ef", type=int, default=0)
    ap.add_argument("--cache-dir", default=os.environ.get("HNSW_BENCH_CACHE", "/tmp/hnsw_mi355x_bench_cache"))
    ap.add_argument("--build-threads", type=int, default=0)
    ap.add_argument("--host-build", action="store_true", help="build the graph on the host cores only (default: GPU-assisted "
                    "construction: the insertions' searches on the device, window by window)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the two-caller-threads reference measurement")
    ap.add_argument("--no-recall", action="store_true", help="skip the brute-force ground truth (quick A/B runs)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child runs behind roofline.traffic (N = 1 only; ~1 minute)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the timings of the host-buffer / reference-FFI / filtered entry points")
    ap.add_argument("--dump-stats", default="", help="write the per-query kernel stats of the last step to this .npy")
    ap.add_argument("--dump-answers", default="", help="write the answers of batch 0 (all ranks' shards gathered, input order) to this .npz")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (gloo only to exercise the multi-rank path on a 1-GPU box)")
    ap.add_argument("--weak", action="store_true", help="N > 1: a fixed number of queries per GPU (12 500 for the SIFT1M shape) "
                    "instead of ONE batch of 100 000 split N ways")
    ap.add_argument("--plan-only", action="store_true", help="N > 1: the ranks only rendezvous (gloo, no GPU needed), agree on who "
                    "searches which block of the batch, and rank 0 prints that plan as JSON (tests/test_sharding.py)")
    ap.add_argument("--share-device", action="store_true", help="all ranks use HIP device 0 (1-GPU box test of the N>1 path)")
    ap.add_argument("--exchange", action="store_true",
                    help="N = 1 only: run the N > 1 exchange anyway -- a ONE-rank process group of --backend (nccl = RCCL) and the same "
                         "overlapped packed all-gather per step -- so that the RCCL path is exercised on a 1-GPU box (rehearsal of the 8-GPU run)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--batches", type=int, default=4, help="distinct query batches the steps rotate through (a step that "
                    "re-searches the batch of the previous step finds its rows in the 256 MiB Infinity Cache)")
    args = ap.parse_args(json.loads(os.environ["HNSW_BENCH_ARGV"]) if "HNSW_BENCH_ARGV" in os.environ and len(sys.argv) == 1 else None)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # the plain command: this process launches one rank per GPU and waits for them

    import torch  # first: the C-ABI library then binds to the HIP runtime torch already loaded
    import hnsw_rs_amd as H

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): pass the same number to both")
    if args.plan_only:
        # no device: every rank works out the plan, the ranks gather the block each of them would search, rank 0 prints it
        import datetime
        import torch.distributed as dist
        from hnsw_rs_amd.sharded import shard_bounds
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
        plan = plan_queries(CONFIGS[args.config], world, args.nq, args.weak)
        lo, hi = shard_bounds(plan["nq_total"], world, rank)
        mine = torch.tensor([lo, hi], dtype=torch.int64)
        blocks = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(blocks, mine)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "config": {"queries_total": plan["nq_total"], "queries_per_gpu": plan["nq_local"]},
                              "scaling": plan["scaling"], "plan": plan["mode"], "queries_dropped": plan["dropped"],
                              "blocks": [[int(b[0]), int(b[1])] for b in blocks]}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    if args.share_device:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: HIP device {local_rank} does not exist ({torch.cuda.device_count()} visible); one rank per GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_pg = None
    rccl = None
    backend_used = args.backend
    pg = world > 1 or args.exchange  # a process group exists and every step ends with the exchange of the packed answers
    if pg:
        import datetime
        import torch.distributed as dist_pg_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:  # --exchange without a launcher: a rendezvous of one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        fallback_reason = os.environ.get("HNSW_BENCH_GATHER_FALLBACK") if args.share_device else None
        if args.backend == "nccl" and fallback_reason is None:
            # backend "nccl" IS RCCL on ROCm; the communicator is created here (device_id)
            dist_pg_mod.init_process_group("nccl", timeout=datetime.timedelta(minutes=30), device_id=dev)
        else:
            # gloo: asked for, or the launcher found that RCCL does not serve ranks sharing one device (--share-device only)
            backend_used = "gloo"
            dist_pg_mod.init_process_group("gloo", timeout=datetime.timedelta(hours=2))
        dist_pg = dist_pg_mod
    coll_dev = dev if backend_used == "nccl" else torch.device("cpu")  # gloo collectives run on host tensors
    if pg:
        seen = torch.ones(1, dtype=torch.int64, device=coll_dev)
        dist_pg.all_reduce(seen)  # every rank of the group adds one: the ranks this communicator really spans
        try:
            rccl_version = "RCCL %s" % ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001 -- informational only
            rccl_version = "RCCL"
        rccl = {"backend": dist_pg.get_backend(), "requested": args.backend, "ranks_seen": int(seen.item()),
                "library": rccl_version if backend_used == "nccl" else "gloo (host tensors)",
                "fallback_reason": fallback_reason, "devices": "all ranks on HIP device 0 (--share-device)" if args.share_device else "one HIP device per rank"}

    cfg = dict(CONFIGS[args.config])
    if args.n:
        cfg["n"] = args.n
    if args.ef:
        cfg["ef"] = args.ef
    plan = plan_queries(cfg, world, args.nq, args.weak)
    nq_local = plan["nq_local"]
    n, d, k, ef = cfg["n"], cfg["d"], cfg["k"], cfg["ef"]
    H.build_native()
    lib = H.lib()

    # ---------------------------------------------------------------- index (cached hnswio dump)
    builder = "host" if args.host_build else "gpu"
    key = hashlib.sha1(json.dumps([args.config, n, d, cfg["dist"], cfg["M"], cfg["efc"], args.data, "v2", builder]).encode()).hexdigest()[:12]
    os.makedirs(args.cache_dir, exist_ok=True)
    base = f"bench_{args.config}_{key}"
    done_marker = os.path.join(args.cache_dir, base + ".done")
    t_build = 0.0
    if rank == 0 and not os.path.exists(done_marker):
        log(f"building {cfg['label']} ({args.data} data), {'host cores only' if args.host_build else 'GPU-assisted'} ...")
        X = synth(n, d, 0x5EED0001, args.data)
        if cfg["dist"] == "DistDot":
            X /= np.linalg.norm(X, axis=1, keepdims=True)
        t0 = time.time()
        hb = H.Hnsw(cfg["M"], n, 16, cfg["efc"], cfg["dist"])
        if args.host_build:
            hb.set_build_options(nthreads=args.build_threads, fast_arithmetic=True)
        else:
            hb.set_build_options(nthreads=args.build_threads, gpu_device=local_rank, gpu_window=0)
        hb.parallel_insert(X)
        t_build = time.time() - t0
        log(f"built in {t_build:.1f} s ({n / t_build:.0f} points/s); dumping to {args.cache_dir}")
        hb.file_dump(args.cache_dir, base)
        del hb, X
        with open(done_marker, "w") as f:
            f.write(json.dumps({"build_s": t_build, "builder": builder}))
    while not os.path.exists(done_marker):  # other ranks: wait on the file system, not on a collective
        time.sleep(1.0)
    t0 = time.time()
    index = H.HnswIo(args.cache_dir, base).load_hnsw(cfg["dist"])
    index.upload(local_rank)
    t_load = time.time() - t0
    log(f"rank {rank}: dump reloaded + uploaded to HBM in {t_load:.1f} s; nb_point={index.get_nb_point()} "
        f"max_level={index.get_max_level_observed()}")

    # ---------------------------------------------------------------- queries (resident in HBM)
    # NB distinct batches; step i searches batch i % NB, so that consecutive steps do not touch the same rows
    nq_total = nq_local * world
    NB = max(1, args.batches)
    Q_all = synth(nq_total * NB, d, 0x5EED0002, args.data)
    if cfg["dist"] == "DistDot":
        Q_all /= np.linalg.norm(Q_all, axis=1, keepdims=True)
    Qh = [np.ascontiguousarray(Q_all[b * nq_total + rank * nq_local: b * nq_total + (rank + 1) * nq_local]) for b in range(NB)]
    Q = Qh[0]
    Qds = [torch.from_numpy(q).to(dev) for q in Qh]
    Qd = Qds[0]
    # the answers of a rank live side by side in ONE byte buffer (ids | distances | counts): the search writes the collective's
    # send buffer in place and the exchange is ONE all-gather per step.  N > 1: two such buffers alternate, so that the exchange
    # of step i (its own stream inside RCCL) overlaps the search of step i + 1; a buffer is rewritten only after its exchange
    # has been waited for, and the timed region ends with every exchange complete (hnsw_rs_amd.sharded.OverlappedExchange;
    # CPU-tested over gloo in tests/test_sharding.py)
    from hnsw_rs_amd.sharded import OverlappedExchange, PackedAnswers
    xch = OverlappedExchange(nq_total, k, world, dev, coll_dev) if pg else None
    packed = xch.packs[0] if xch else PackedAnswers(nq_local, k, dev)
    out_ids, out_dists, out_counts = packed.ids, packed.dists, packed.counts
    out_layer = torch.zeros((nq_local, k), dtype=torch.uint8, device=dev)
    out_rank = torch.zeros((nq_local, k), dtype=torch.int32, device=dev)
    stats = torch.zeros((nq_local, 8), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    kernel_ms = []
    main_ms = []
    gather_marks = []  # per step: (event before, event after) the all-gather on the launch stream, or host seconds (gloo)

    def drain_exchanges():
        if xch:
            xch.drain()

    def step(i, overlap=True, exchange=True):
        pk = xch.buffer(i) if xch else packed  # (N > 1: waits for the exchange that last read this buffer, two steps ago)
        rc = lib.hnswgpu_search_batch_device(index.handle, Qds[i % NB].data_ptr(), nq_local, d, k, ef, pk.ids.data_ptr(),
                                             pk.dists.data_ptr(), out_layer.data_ptr(), out_rank.data_ptr(),
                                             pk.counts.data_ptr(), stats.data_ptr(), stream.cuda_stream)
        if rc != 0:
            raise RuntimeError(H._native.last_error())
        ms, _ = index.last_kernel_ms()  # HIP events on the launch stream, inside the library
        kernel_ms.append(ms)
        main_ms.append(index.last_search_kernel_ms())
        if pg and exchange:  # the only exchange on this path: ONE all-gather of the packed answers (RCCL over xGMI)
            t0 = time.perf_counter()
            xch.exchange(i, overlap)
            if not overlap:  # (measured alone: what an exchange costs when nothing hides it)
                if backend_used == "nccl":
                    torch.cuda.synchronize(dev)
                gather_marks.append(time.perf_counter() - t0)

    def fence():
        drain_exchanges()
        if pg:
            dist_pg.barrier()
        torch.cuda.synchronize(dev)

    def timed(nsteps):
        fence()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(i)
        fence()
        el = time.perf_counter() - t0
        if pg:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist_pg.all_reduce(t, op=dist_pg.ReduceOp.MAX)
            el = float(t.item())
        return el

    # The device idles through the setup above (index load, data generation on the host cores) and its clocks fall back: the
    # first ~0.1 s of work after an idle period runs several times slower (measured: 27 ms instead of 1.4 ms per call right after
    # a 5 s pause).  Untimed spin-up until the clocks are back, then the W warm-up steps the contract asks for, then the K timed ones.
    # (no exchange in there: the ranks spin by the clock, not by a common count, and collectives must pair up)
    _spin = [0]

    def _spin_step():  # (the batches rotate here too: a profiler's per-kernel average is then over all of them)
        step(_spin[0], exchange=False)
        _spin[0] += 1
    spin_up(_spin_step)
    for i in range(args.warmup):
        step(i)
    kernel_ms.clear()
    main_ms.clear()
    gather_marks.clear()
    elapsed = timed(args.steps)
    ms_per_step = elapsed * 1e3 / args.steps
    qps = nq_total * args.steps / elapsed
    timed_main_ms = list(main_ms)
    timed_kernel_ms = list(kernel_ms)
    gather_ms = None
    if pg:  # untimed: a few steps with the exchange NOT overlapped and waited for, to say what one costs by itself
        gather_marks.clear()
        for i in range(4):
            step(i, overlap=False)
        fence()
        gather_ms = float(np.median(gather_marks)) * 1e3
        kernel_ms[:] = kernel_ms[:len(timed_kernel_ms)]
        main_ms[:] = main_ms[:len(timed_main_ms)]

    # N > 1 (one batch shared by the GPUs): the same batch -- all nq_total queries -- in ONE call on rank 0's GPU, untimed by the
    # contract's clock: what a single GPU makes of BASELINE.json configs[3], on this box, this graph, this invocation
    one_gpu_qps = None
    if world > 1 and plan["scaling"] == "strong":
        fence()
        if rank == 0:
            Qw = torch.from_numpy(np.ascontiguousarray(Q_all[:nq_total])).to(dev)
            w_ids = torch.zeros((nq_total, k), dtype=torch.int64, device=dev)
            w_d = torch.zeros((nq_total, k), dtype=torch.float32, device=dev)
            w_cnt = torch.zeros((nq_total,), dtype=torch.int32, device=dev)
            w_layer = torch.zeros((nq_total, k), dtype=torch.uint8, device=dev)
            w_rank = torch.zeros((nq_total, k), dtype=torch.int32, device=dev)
            w_stats = torch.zeros((nq_total, 8), dtype=torch.int32, device=dev)

            def whole():
                rc = lib.hnswgpu_search_batch_device(index.handle, Qw.data_ptr(), nq_total, d, k, ef, w_ids.data_ptr(), w_d.data_ptr(),
                                                     w_layer.data_ptr(), w_rank.data_ptr(), w_cnt.data_ptr(), w_stats.data_ptr(), stream.cuda_stream)
                if rc != 0:
                    raise RuntimeError(H._native.last_error())
            whole()
            torch.cuda.synchronize(dev)
            reps = max(2, min(args.steps, 5))
            t0 = time.perf_counter()
            for _ in range(reps):
                whole()
            torch.cuda.synchronize(dev)
            one_gpu_qps = nq_total * reps / (time.perf_counter() - t0)
            del Qw, w_ids, w_d, w_cnt, w_layer, w_rank, w_stats
        fence()

    # untimed accounting: one more step per batch for its work counters (the algorithmic bytes of that batch)
    batch_stats = []
    for b in range(NB):
        step(b)
        fence()
        batch_stats.append(stats.cpu().numpy().astype(np.int64))

    # for reference: the same steps with strict ties OFF (equal distances ordered by arrival, queries flagged)
    fast_qps = None
    if lib.hnswgpu_set_strict_ties(index.handle, 0) == 0:
        step(0)
        fast_qps = nq_total * args.steps / timed(args.steps)
        lib.hnswgpu_set_strict_ties(index.handle, 1)
    # for reference: the same steps (strict) with the distances summed in the order of the crate's simdeez_f build -- the
    # arithmetic of every number the reference publishes, and the fair twin of the CPU baseline's "simd-order" figure
    # (hnswgpu_set_arithmetic; opt-in, never `value`; its checker is the oracle's dist_simd8: tests/test_gpu_round4.py)
    simd_qps = None
    if lib.hnswgpu_set_arithmetic(index.handle, 1) == 0:
        try:
            step(0)
            simd_qps = nq_total * args.steps / timed(args.steps)
        finally:
            lib.hnswgpu_set_arithmetic(index.handle, 0)
    # for reference: the same steps issued by two caller threads (each its own stream and output buffers; the library
    # serves concurrent calls from a workspace pool).  One launch ends with its longest search while the machine
    # drains; a second batch in flight fills that tail.  Not `value`: one caller, one batch at a time.
    two_callers_qps = None
    tickets_qps = None
    if world == 1 and not args.no_concurrent:
        import threading
        lanes = []
        for _ in range(2):
            lanes.append({"stream": torch.cuda.Stream(dev),
                          "ids": torch.zeros_like(out_ids), "dists": torch.zeros_like(out_dists), "layer": torch.zeros_like(out_layer),
                          "rank": torch.zeros_like(out_rank), "counts": torch.zeros_like(out_counts), "stats": torch.zeros_like(stats)})
        errors = []

        def caller(li, nsteps):
            ln = lanes[li]
            for i in range(li, nsteps, 2):
                rc = lib.hnswgpu_search_batch_device(index.handle, Qds[i % NB].data_ptr(), nq_local, d, k, ef, ln["ids"].data_ptr(),
                                                     ln["dists"].data_ptr(), ln["layer"].data_ptr(), ln["rank"].data_ptr(),
                                                     ln["counts"].data_ptr(), ln["stats"].data_ptr(), ln["stream"].cuda_stream)
                if rc != 0:
                    errors.append(H._native.last_error())
                    return

        def run_callers(nsteps):
            fence()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=caller, args=(li, nsteps), daemon=True) for li in range(2)]
            for t in ths:
                t.start()
            for t in ths:
                t.join(timeout=300.0)
            if any(t.is_alive() for t in ths):
                raise RuntimeError("a caller thread did not come back")
            fence()
            return time.perf_counter() - t0

        try:  # an informational figure: never at the expense of the line itself
            run_callers(2 * max(1, args.warmup))
            el2 = run_callers(args.steps)
            if errors:
                raise RuntimeError(errors[0])
            two_callers_qps = nq_total * args.steps / el2
        except Exception as e:  # noqa: BLE001
            log(f"two-caller measurement skipped: {e}")
            two_callers_qps = None
        # the same overlap from ONE host thread through the library's asynchronous calls (hnswgpu_search_batch_device_begin /
        # hnswgpu_search_batch_end: a ticket per batch, two in flight) -- what a Rust host would do instead of a second thread
        try:
            def run_tickets(nsteps):
                fence()
                t0 = time.perf_counter()
                pending = []
                for i in range(nsteps):
                    ln = lanes[i % 2]
                    if len(pending) == 2:
                        if lib.hnswgpu_search_batch_end(pending.pop(0)) != 0:
                            raise RuntimeError(H._native.last_error())
                    tk = C.c_void_p()
                    rc = lib.hnswgpu_search_batch_device_begin(index.handle, Qds[i % NB].data_ptr(), nq_local, d, k, ef, ln["ids"].data_ptr(),
                                                               ln["dists"].data_ptr(), ln["layer"].data_ptr(), ln["rank"].data_ptr(),
                                                               ln["counts"].data_ptr(), ln["stats"].data_ptr(), ln["stream"].cuda_stream, C.byref(tk))
                    if rc != 0:
                        raise RuntimeError(H._native.last_error())
                    pending.append(tk)
                for tk in pending:
                    if lib.hnswgpu_search_batch_end(tk) != 0:
                        raise RuntimeError(H._native.last_error())
                fence()
                return time.perf_counter() - t0
            run_tickets(2 * max(1, args.warmup))
            tickets_qps = nq_total * args.steps / run_tickets(args.steps)
        except Exception as e:  # noqa: BLE001
            log(f"two-ticket measurement skipped: {e}")
            tickets_qps = None
    boundary = None
    orc = None  # the oracle: checker and CPU baseline (rank 0, N = 1 only), never on the product path
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib
        t0 = time.time()
        orc = oracle_lib.OracleHnsw.load(args.cache_dir, base, cfg["dist"])
        log(f"oracle reloaded the same dump in {time.time() - t0:.1f} s")
    if world == 1 and not args.no_boundary:
        try:  # informational figures: never at the expense of the line itself
            boundary = boundary_timings(H, lib, index, args.cache_dir, base, cfg["dist"], Q, k, ef, n)
            # row f3, measured like the main row: counters -> algorithmic bytes -> fraction of the peak, the oracle's
            # search_filter beside it, parity of its sample; at 2 000 queries per call (what rounds 2-3 reported) and at the
            # main row's batch size
            boundary["filtered"] = {"note": "Hnsw::search_filter with a sorted id vector allowing 1 % / 30 % of the points, device-resident buffers "
                                            "(hnswgpu_search_batch_filtered_device), allow bitmap built per call, hnsw_search_exact_kernel",
                                    "2000_queries_per_call": filtered_measure(torch, H, lib, index, orc, n, d, k, ef, Q[:2000], 128, [1, 30]),
                                    "%d_queries_per_call" % len(Q): filtered_measure(torch, H, lib, index, None, n, d, k, ef, Q, 0, [1, 30])}
        except Exception as e:  # noqa: BLE001
            log(f"boundary timings skipped: {e}")
    step(0)   # leave the strict answers of batch 0 in the output buffers for the recall / parity checks below
    fence()
    if args.dump_answers and rank == 0:  # batch 0 in input order: what the caller of parallel_search gets back
        if pg:  # (what came back through the collective, not this rank's send buffer)
            g_ids, g_dists, g_counts = xch.gathered(0)
            np.savez(args.dump_answers, ids=g_ids.cpu().numpy(), dists=g_dists.cpu().numpy(), counts=g_counts.cpu().numpy())
        else:
            np.savez(args.dump_answers, ids=out_ids.cpu().numpy(), dists=out_dists.cpu().numpy(), counts=out_counts.cpu().numpy())

    # ---------------------------------------------------------------- recall vs exact brute force
    res_ids = out_ids.cpu().numpy()
    res_d = out_dists.cpu().numpy().astype(np.float64)
    cnt = out_counts.cpu().numpy()
    hit_id = 0
    hit_dist = 0
    if not args.no_recall:
        Xd = torch.empty((n, d), dtype=torch.float32, device=dev)
        # the same vectors the index was built from (regenerated, not read back through the API)
        Xh = synth(n, d, 0x5EED0001, args.data)
        if cfg["dist"] == "DistDot":
            Xh /= np.linalg.norm(Xh, axis=1, keepdims=True)
        Xd.copy_(torch.from_numpy(Xh))
        del Xh
        gt_ids, gt_d = ground_truth(torch, Xd, Qd, k, cfg["dist"])
        del Xd
        gt_ids_h, gt_d_h = gt_ids.cpu().numpy(), gt_d.cpu().numpy()
        for i in range(nq_local):
            c = int(cnt[i])
            hit_id += len(set(res_ids[i, :c].tolist()) & set(gt_ids_h[i].tolist()))
            hit_dist += int((res_d[i, :c] <= gt_d_h[i, k - 1] * (1 + 1e-6)).sum())  # examples/ann-sift1m...:172-186
    recall = np.array([hit_id, hit_dist, nq_local * k], dtype=np.float64)
    if world > 1:
        t = torch.from_numpy(recall).to(coll_dev)
        dist_pg.all_reduce(t)
        recall = t.cpu().numpy()
    recall_id, recall_dist = recall[0] / recall[2], recall[1] / recall[2]

    # ---------------------------------------------------------------- roofline of the search kernel
    st = stats.cpu().numpy().astype(np.int64)   # batch 0, strict
    if args.dump_stats and rank == 0:
        np.save(args.dump_stats, st)

    def alg_bytes_of(sb, part="search"):
        # bytes = n_dist*d*4 + n_ids_read*4 + n_expand*8 + d*4 + k*12 per query, d unpadded.  The greedy descent
        # of a query runs in hnsw_descend_kernel, in front of the search kernel: stats word 7 says what of the counters is
        # its share (lists scanned << 8 | n_dist << 16), and the dominant kernel is priced on its own share only.
        nd_desc, nx_desc = (sb[:, 7] >> 16) & 0xFFFF, (sb[:, 7] >> 8) & 0xFF
        desc = int(nd_desc.sum()) * d * 4 + int((nd_desc - 1).clip(min=0).sum()) * 4 + int(nx_desc.sum()) * 8 + nq_local * d * 4
        if part == "descent":
            return desc
        total = int(sb[:, 0].sum()) * d * 4 + int(sb[:, 2].sum()) * 4 + int(sb[:, 1].sum()) * 8 + nq_local * (d * 4 + k * 12)
        return total - (desc - nq_local * d * 4)  # (the search kernel stages the query row as well)

    batch_bytes = [alg_bytes_of(sb) for sb in batch_stats]
    descent_bytes = float(np.mean([alg_bytes_of(sb, "descent") for sb in batch_stats]))
    n_dist, n_expand, n_ids = (int(sum(sb[:, c].sum() for sb in batch_stats)) / NB for c in (0, 1, 2))
    # the dominant kernel is hnsw_search_kernel (queries that need the literal heaps carry on inside it); achieved =
    # algorithmic bytes of the batches the timed steps searched / the kernel time of those launches (HIP events)
    k_ms = float(np.mean(timed_main_ms)) if timed_main_ms else float("nan")
    all_ms = float(np.mean(timed_kernel_ms)) if timed_kernel_ms else float("nan")
    alg_bytes = float(np.mean([batch_bytes[i % NB] for i in range(args.steps)]))
    achieved = sum(batch_bytes[i % NB] for i in range(args.steps)) / (sum(timed_main_ms) * 1e-3) / 1e9
    # A call whose first pass hands queries on (two queries per wavefront first, then the one-query kernels over the tie queries:
    # several search launches per step) is priced over ALL kernels of the call -- the second launch's bytes are in the count, so
    # its time (and, conservatively, the descent's and the host's turn-around between the launches) is too.
    several_launches = index.last_kernel_ms()[1] > 1
    if several_launches:
        achieved = sum(batch_bytes[i % NB] for i in range(args.steps)) / (sum(timed_kernel_ms) * 1e-3) / 1e9
    traffic_profile = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            ent = tj.get(args.config)
            if ent and ent.get("data") == args.data:
                traffic_profile = ent
        except Exception:
            pass
    index_bytes = n * (((d + 31) // 32) * 128 + 4 * ((2 * cfg["M"] + 15) // 16) * 16 + 8)
    all_st = np.concatenate(batch_stats)
    roofline = {"bound": "hbm", "kernel": "hnsw_search_kernel" if not several_launches else
                "all search launches of a call (hnsw_search_pair_kernel first, hnsw_search_kernel over what it handed on): achieved = algorithmic bytes / all_kernels_ms",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                # PMC counters cannot be collected inside this process: `traffic` is filled in below from two rocprofv3 --pmc
                # child runs (live_traffic; N = 1), and the committed profile of an earlier box is quoted beside it
                "traffic": None, "traffic_from_profile": traffic_profile,
                "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": round(k_ms, 4),
                "all_kernels_ms": round(all_ms, 4),
                # in front of the search kernel on the same stream: hnsw_descend_kernel (queries padded + the greedy descent of
                # every query, exact arithmetic) and order_desc_kernel (longest searches first)
                "descent_and_order_kernels": {"ms": round(all_ms - k_ms, 4), "algorithmic_bytes_per_launch": int(descent_bytes),
                                              "note": "their bytes and their time are NOT in achieved / kernel_ms: the dominant kernel is priced on its own share"},
                "queries_resolved_with_literal_heaps": int((st[:, 3] == 3).sum()),
                "queries_that_met_equal_distances": int((st[:, 7] & 1).sum()),
                "launches_per_step": index.last_kernel_ms()[1], "query_batches_rotated": NB,
                # the batches differ (their longest search sets a launch's length): spread of the timed launches per batch
                "kernel_ms_by_batch_min_median_max": {str(b): [round(float(f(v)), 4) for f in (np.min, np.median, np.max)]
                                                      for b in range(NB) for v in [[m for i, m in enumerate(timed_main_ms) if i % NB == b]] if v},
                "index_bytes_in_hbm": int(index_bytes),
                "resident_in_infinity_cache": bool(index_bytes < 256 * 2 ** 20),
                "note": ("the whole index fits the 256 MiB Infinity Cache: the fetches behind `achieved` are served by MALL/L2, "
                         "the fraction is against the HBM peak only by convention" if index_bytes < 256 * 2 ** 20 else
                         "working set (vectors + lists) exceeds the Infinity Cache; batches rotate so that steps do not re-read each other's rows"),
                "per_query": {"n_dist": n_dist / nq_local, "n_expand": n_expand / nq_local,
                              "n_ids_read": n_ids / nq_local, "bytes": alg_bytes / nq_local,
                              "n_dist_p50_p99_max": [int(np.percentile(all_st[:, 0], 50)), int(np.percentile(all_st[:, 0], 99)), int(all_st[:, 0].max())]}}

    # ---------------------------------------------------------------- CPU baseline (oracle = checker)
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, parity = cpu_baseline_leg(args.cache_dir, base, cfg["dist"], Q, k, ef, res_ids, out_dists.cpu().numpy(),
                                                st, cnt, args.cpu_seconds, orc)

    if cpu_baseline is not None:
        # the device's figure in the SAME arithmetic as cpu_baseline.value (which is the better of the port's two orders):
        # `value` itself is always the scalar order, the one the parity block checks
        simd = cpu_baseline.get("arithmetic", "").startswith("simd")
        cpu_baseline["device_queries_per_s_in_the_same_arithmetic"] = (None if simd_qps is None else round(simd_qps, 1)) if simd else round(qps, 1)
        cpu_baseline["scalar_order_value"] = max(cpu_baseline["by_threads"].values())
    if rank == 0:
        out = {
            "metric": "queries/sec (+ recall@10), batched HNSW search",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": plan["scaling"], "vs_baseline": None,
            "dtype": "f32", "data": f"synthetic ({args.data}, seeds 0x5EED0001/0x5EED0002), graph built by the product builder ({'host cores' if args.host_build else 'GPU-assisted construction'})",
            "config": {"workload": cfg["label"], "n": n, "d": d, "distance": cfg["dist"], "M": cfg["M"],
                       "ef_construction": cfg["efc"], "ef": ef, "k": k, "queries_per_gpu": nq_local,
                       "queries_total": nq_total, "query_plan": plan["mode"], "graph": "replicated per GPU",
                       "exchange": ("one all_gather_into_tensor per step of the packed answers (ids | distances | counts, %d bytes per rank) over %s, "
                                    "issued asynchronously: it overlaps the search of the next step (two answer buffers alternate); the timed "
                                    "region ends with every exchange complete" % (xch.shard_bytes, "RCCL" if backend_used == "nccl" else "gloo")) if pg else "none",
                       "parallelism": f"{world} x (replica + {nq_local} queries)"},
            "rccl": rccl,
            "gather_ms": None if gather_ms is None else round(gather_ms, 4),
            # N > 1, strong scaling: the WHOLE batch (queries_total) in one call on rank 0's GPU, same invocation, same graph:
            # value / this = the speed-up of N GPUs over one on BASELINE.json configs[3]
            "one_gpu_same_batch_queries_per_s": None if one_gpu_qps is None else round(one_gpu_qps, 1),
            "speedup_over_one_gpu_same_batch": None if one_gpu_qps is None else round(qps / one_gpu_qps, 3),
            "recall_at_10": None if args.no_recall else {"by_id": round(float(recall_id), 4), "by_distance_threshold": round(float(recall_dist), 4)},
            "strict_ties": {"on": True, "note": "queries whose answer depends on the internal order of the reference's BinaryHeaps (equal f32 distances at a decisive place) carry on with a literal emulation of the heap in question (DESIGN.md section 6)",
                            "fast_mode_queries_per_s": None if fast_qps is None else round(fast_qps, 1)},
            "simd_order_queries_per_s": None if simd_qps is None else round(simd_qps, 1),
            "two_caller_threads_queries_per_s": None if two_callers_qps is None else round(two_callers_qps, 1),
            "one_caller_two_tickets_queries_per_s": None if tickets_qps is None else round(tickets_qps, 1),
            # `value` is the device-resident call (queries already in HBM when the timed region starts, answers left in HBM: the
            # measurement contract of this build).  The reference's API is host buffers (src/libext.rs:205-254, BASELINE.md section 2:
            # "one batched call including host->device query copy and device->host result copy"): that call, same box, same batch:
            "reference_call_queries_per_s": None if not boundary else boundary.get("ffi_parallel_search_neighbours_f32_queries_per_s"),
            "host_buffers_queries_per_s": None if not boundary else boundary.get("host_buffers_queries_per_s"),
            "value_is": "hnswgpu_search_batch_device (inputs and outputs resident in HBM); reference_call_queries_per_s = the reference's own "
                        "symbol parallel_search_neighbours_f32 (row pointers in host memory -> Vec_api<Neighbourhood_api> in host memory, PCIe both ways)",
            "boundary": boundary,
            "roofline": roofline,  # (+ traffic, below)
            "cpu_baseline": cpu_baseline,
            "parity_vs_oracle": parity,
            "setup_s": {"build": round(t_build, 1), "load_upload": round(t_load, 1)},
        }
        if world == 1 and not args.no_traffic:
            # last, so that nothing of it can disturb what was measured above: the device is idle, this process only waits
            t0 = time.time()
            nbytes, detail = live_traffic(args)
            detail["seconds"] = round(time.time() - t0, 1)
            roofline["traffic"] = nbytes
            roofline["traffic_unit"] = "bytes per launch (like algorithmic_bytes_per_launch)"
            roofline["traffic_live"] = detail
            if nbytes:
                roofline["traffic_over_algorithmic"] = round(nbytes / roofline["algorithmic_bytes_per_launch"], 3)
            log(f"roofline.traffic: {nbytes} bytes per launch ({detail.get('seconds')} s)" if nbytes else f"roofline.traffic skipped: {detail.get('skipped')}")
        print(json.dumps(out), flush=True)
    if pg:
        dist_pg.barrier()
        dist_pg.destroy_process_group()


if __name__ == "__main__":
    main()
