#!/usr/bin/env python3
"""Filtered search (row f3) measured: per-query work counters and durations of the device path (stats of
hnswgpu_search_batch_filtered_device) at 1 % / 30 % of the points allowed on the cached bench index of a config, next to the
oracle's search_filter on the same queries (CPU baseline of the row) and a parity check of the answers.
    python tools/filtered_stats.py --config sift1m [--nq 2000] [--cpu-queries 256]"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="sift1m")
ap.add_argument("--nq", type=int, default=2000)
ap.add_argument("--cpu-queries", type=int, default=256)
ap.add_argument("--pcts", default="1,30")
ap.add_argument("--cache-dir", default=os.environ.get("HNSW_BENCH_CACHE", "/tmp/hnsw_mi355x_bench_cache"))
args = ap.parse_args()
import torch  # noqa: E402  (first: see INTEGRATION.md, loading order)
import hnsw_rs_amd as H  # noqa: E402
import oracle_lib  # noqa: E402

cfg = bench.CONFIGS[args.config]
marks = sorted(f for f in glob.glob(os.path.join(args.cache_dir, f"bench_{args.config}_*.done"))
               if len(os.path.basename(f)) == len(f"bench_{args.config}_") + 12 + 5)
if not marks:
    raise SystemExit("run bench.py for this config first (it builds and caches the index)")
base = os.path.basename(marks[-1])[:-5]
index = H.HnswIo(args.cache_dir, base).load_hnsw(cfg["dist"])
index.upload(0)
lib = H.lib()
orc = oracle_lib.OracleHnsw.load(args.cache_dir, base, cfg["dist"])
n, d, k, ef = cfg["n"], cfg["d"], cfg["k"], cfg["ef"]
Q = bench.synth(args.nq, d, 0x5EED0002, "clustered")
if cfg["dist"] == "DistDot":
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
out = bench.filtered_measure(torch, H, lib, index, orc, n, d, k, ef, Q, args.cpu_queries, [int(p) for p in args.pcts.split(",")])
print(json.dumps(out, indent=1))
