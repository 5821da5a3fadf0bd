#!/usr/bin/env python3
"""Filtered search (Hnsw::search_filter with a sorted id vector, hnsw_search_exact_kernel) on the cached bench index of a
config, for a profiler: `rocprofv3 --kernel-trace --stats -- python tools/filtered_run.py --config sift1m`.  Allows 1 % and
30 % of the points, 2 000 queries, three calls each."""
import argparse
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="sift1m")
ap.add_argument("--cache-dir", default=os.environ.get("HNSW_BENCH_CACHE", "/tmp/hnsw_mi355x_bench_cache"))
args = ap.parse_args()
import torch  # noqa: E402,F401  (first: see INTEGRATION.md, loading order)
import hnsw_rs_amd as H  # noqa: E402

cfg = bench.CONFIGS[args.config]
marks = sorted(f for f in glob.glob(os.path.join(args.cache_dir, f"bench_{args.config}_*.done"))
               if len(os.path.basename(f)) == len(f"bench_{args.config}_") + 12 + 5)  # (12 hex digits: not glove25_dot for glove25)
if not marks:
    raise SystemExit("run bench.py for this config first (it builds and caches the index)")
base = os.path.basename(marks[-1])[:-5]
index = H.HnswIo(args.cache_dir, base).load_hnsw(cfg["dist"])
index.upload(0)
Q = bench.synth(2000, cfg["d"], 0x5EED0002, "clustered")
rng = np.random.default_rng(0xF117)
for pct in (1, 30):
    allowed = np.sort(rng.choice(cfg["n"], max(1, cfg["n"] * pct // 100), replace=False)).astype(np.uint64)
    for _ in range(3):
        r = index.parallel_search_filter_flat(Q, cfg["k"], cfg["ef"], allowed)
    print(f"{pct} % allowed: kernels of the last call {index.last_kernel_ms()[0]:.2f} ms, answers with k entries: {(r.counts == cfg['k']).mean():.3f}")
