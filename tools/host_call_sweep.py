#!/usr/bin/env python3
"""The host-buffer boundary of one config under several settings of the library's tuning hooks, in ONE process on ONE index (graphs
differ per box): hnswgpu_search_batch with output arrays that are reused (what a caller in steady state does -- fresh numpy arrays
take a page fault per 4 KB while the answers are unpacked), the reference's FFI symbol, and the device-resident call beside them.
    python bench.py --config sift1m --steps 2 --warmup 1 --no-boundary --no-cpu-baseline --no-recall --no-traffic   # builds and caches the index
    python tools/host_call_sweep.py --config sift1m "HNSWGPU_HOST_THREADS=16" "HNSWGPU_HOST_THREADS=16,HNSWGPU_HOST_CHUNKS=8"
"""
import argparse
import ctypes as C
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="sift1m")
ap.add_argument("--cache-dir", default=os.environ.get("HNSW_BENCH_CACHE", "/tmp/hnsw_mi355x_bench_cache"))
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--build-first", type=int, default=0, help="build (and drop) an index of this many points in the process first, GPU-assisted, "
                "on every core: what bench.py's process has behind it when it measures the boundary (hundreds of pool threads)")
ap.add_argument("settings", nargs="*", help="ENV=V[,ENV=V] per setting; the default environment is measured first and last")
args = ap.parse_args()
import torch  # noqa: E402
import hnsw_rs_amd as H  # noqa: E402

cfg = bench.CONFIGS[args.config]
if args.build_first:
    hb = H.Hnsw(cfg["M"], args.build_first, 16, cfg["efc"], cfg["dist"])
    hb.set_build_options(nthreads=0, gpu_device=0, gpu_window=0)
    hb.parallel_insert(bench.synth(args.build_first, cfg["d"], 0x5EED0009, "clustered"))
    del hb
marks = sorted(f for f in glob.glob(os.path.join(args.cache_dir, f"bench_{args.config}_*.done"))
               if len(os.path.basename(f)) == len(f"bench_{args.config}_") + 12 + 5)
if not marks:
    raise SystemExit("run bench.py for this config first (it builds and caches the index)")
base = os.path.basename(marks[-1])[:-5]
index = H.HnswIo(args.cache_dir, base).load_hnsw(cfg["dist"])
index.upload(0)
lib = H.lib()
nq, d, k, ef = cfg["nq"], cfg["d"], cfg["k"], cfg["ef"]
Q = bench.synth(nq, d, 0x5EED0002, "clustered")
ids = np.zeros((nq, k), np.uint64); dists = np.zeros((nq, k), np.float32); layers = np.zeros((nq, k), np.uint8)
ranks = np.zeros((nq, k), np.int32); counts = np.zeros(nq, np.uint32)
p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


def host_call():
    rc = lib.hnswgpu_search_batch(index.handle, p(Q), nq, d, k, ef, p(ids), p(dists), p(layers), p(ranks), p(counts))
    assert rc == 0, H._native.last_error()


def fresh_arrays_call():
    index.parallel_search_flat(Q, k, ef)


Qd = torch.from_numpy(Q).cuda()
t_ids = torch.zeros((nq, k), dtype=torch.int64, device="cuda"); t_d = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
t_l = torch.zeros((nq, k), dtype=torch.uint8, device="cuda"); t_r = torch.zeros((nq, k), dtype=torch.int32, device="cuda")
t_c = torch.zeros((nq,), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream


def device_call():
    rc = lib.hnswgpu_search_batch_device(index.handle, Qd.data_ptr(), nq, d, k, ef, t_ids.data_ptr(), t_d.data_ptr(), t_l.data_ptr(),
                                         t_r.data_ptr(), t_c.data_ptr(), None, stream)
    assert rc == 0, H._native.last_error()
    torch.cuda.synchronize()


api = None
loader = getattr(lib, "load_hnswdump_f32_" + cfg["dist"], None)
if loader is not None:
    cwd = os.getcwd()
    os.chdir(args.cache_dir)
    try:
        api = loader(lib.get_hnswio(len(base), base.encode()))
    finally:
        os.chdir(cwd)
rows = (C.c_void_p * nq)(*[Q.ctypes.data + i * d * 4 for i in range(nq)])


def ffi_call():
    v = lib.parallel_search_neighbours_f32(api, nq, d, rows, k, ef)
    assert v, H._native.last_error()
    lib.hnswgpu_free_neighbourhood_vec(v)


def rate(fn):
    bench.spin_up(fn)
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3


def measure(tag):
    row = {"device": rate(device_call), "host_reused": rate(host_call), "host_fresh": rate(fresh_arrays_call)}
    if api:
        row["ffi"] = rate(ffi_call)
    print(f"{tag:50s} " + "  ".join(f"{n} {ms:.3f} ms ({nq / ms / 1e3:.2f} M/s)" for n, ms in row.items()), flush=True)


measure("default")
for s in args.settings:
    kv = dict(x.split("=", 1) for x in s.split(","))
    for a, b in kv.items():
        os.environ[a] = b
    measure(s)
    for a in kv:
        del os.environ[a]
measure("default again")
