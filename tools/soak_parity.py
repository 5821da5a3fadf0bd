#!/usr/bin/env python3
"""Soak test at full size: many fresh query batches against the graph a bench run left in its cache, every answer of the
strict path compared with the oracle (ids, f32 distance bits, p_ids, counts).  Run after `python bench.py --config C`
on the same box:   python tools/soak_parity.py --config sift1m --batches 10
(The oracle is the checker here, like in tests/ -- this is a test driver, not a product path.)"""
import argparse
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402  (CONFIGS, synth)
import hnsw_rs_amd as H  # noqa: E402
import oracle_lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="sift1m", choices=sorted(bench.CONFIGS))
ap.add_argument("--batches", type=int, default=8)
ap.add_argument("--seed-base", type=lambda v: int(v, 0), default=0xABCD0000, help="batch b is drawn with seed base + b")
ap.add_argument("--points-as-queries", type=int, default=0, help="this many queries of every batch are stored points (distance 0, duplicates' ties)")
ap.add_argument("--simd8", action="store_true", help="both sides in the SIMD summation order (hnswgpu_set_arithmetic / the oracle's set_simd_order)")
ap.add_argument("--filtered", type=int, default=0, metavar="N", help="additionally, per batch: the first N queries through search_filter "
                "with a fresh random filter allowing 1 % (even batches) or 30 % (odd batches) of the points")
ap.add_argument("--cache-dir", default=os.environ.get("HNSW_BENCH_CACHE", "/tmp/hnsw_mi355x_bench_cache"))
args = ap.parse_args()
cfg = bench.CONFIGS[args.config]
dumps = sorted(f for f in glob.glob(os.path.join(args.cache_dir, f"bench_{args.config}_*.done"))
               if len(os.path.basename(f)) == len(f"bench_{args.config}_") + 12 + 5)  # (12 hex digits: not glove25_dot for glove25)
if not dumps:
    sys.exit(f"no cached graph for {args.config} in {args.cache_dir}: run bench.py --config {args.config} first")
base = os.path.basename(dumps[-1])[:-5]
h = H.HnswIo(args.cache_dir, base).load_hnsw(cfg["dist"])
h.upload(0)
o = oracle_lib.OracleHnsw.load(args.cache_dir, base, cfg["dist"])
k, ef, d, nq = cfg["k"], cfg["ef"], cfg["d"], cfg["nq"]
dm = H.DataMap.from_hnswdump(args.cache_dir, base) if args.points_as_queries else None
if args.simd8:
    h.set_arithmetic("simd8")
    o.set_simd_order(True)
bad = checked = ties = fbad = fchecked = 0
t0 = time.time()
for b in range(args.batches):
    Q = bench.synth(nq, d, args.seed_base + b, "clustered" if b % 2 == 0 else "uniform")
    if args.points_as_queries:
        pick = np.random.default_rng(args.seed_base + b).choice(h.get_nb_point(), args.points_as_queries, replace=False)
        Q[: args.points_as_queries] = np.stack([dm.get_data(int(i)) for i in pick])  # (origin ids of a bench graph are 0 .. n-1)
    if cfg["dist"] == "DistDot":
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    res = h.parallel_search_flat(Q, k, ef)
    ties += h.last_tie_count()
    ref = o.parallel_search(Q, k, ef)
    ok = (np.all(res.ids == ref.ids, axis=1) & np.all(res.dists.view(np.uint32) == ref.dists.view(np.uint32), axis=1)
          & np.all(res.layers == ref.layers, axis=1) & np.all(res.ranks == ref.ranks, axis=1) & (res.counts == ref.counts))
    bad += int((~ok).sum())
    checked += nq
    if args.filtered:
        nf = min(nq, args.filtered)
        n = h.get_nb_point()
        pct = 1 if b % 2 == 0 else 30
        allowed = np.sort(np.random.default_rng(args.seed_base + 77 + b).choice(n, max(1, n * pct // 100), replace=False)).astype(np.uint64)
        fr = h.parallel_search_filter_flat(Q[:nf], k, ef, allowed)
        fo = o.parallel_search_filter(Q[:nf], k, ef, allowed)
        fok = (fr.counts == fo.counts)
        for i in range(nf):
            c = int(fo.counts[i])
            fok[i] = fok[i] and np.array_equal(fr.ids[i, :c], fo.ids[i, :c]) and np.array_equal(fr.dists[i, :c].view(np.uint32), fo.dists[i, :c].view(np.uint32))
        fbad += int((~fok).sum())
        fchecked += nf
        for i in np.nonzero(~fok)[0][:4]:
            print(f"  batch {b} ({pct} % allowed, filter seed {args.seed_base + 77 + b:#x}), query {i}: counts {fr.counts[i]} / {fo.counts[i]}, status {fr.status[i]} / {fo.status[i]}\n"
                  f"    device ids {fr.ids[i].tolist()}\n    oracle ids {fo.ids[i].tolist()}\n    device d {fr.dists[i].tolist()}\n    oracle d {fo.dists[i].tolist()}")
if args.filtered:
    print(f"{args.config}: {fchecked} filtered queries (1 % / 30 % allowed alternating), {fbad} differ from the oracle")
    bad += fbad
print(f"{args.config}{' (SIMD-order arithmetic)' if args.simd8 else ''}: {checked} queries in {args.batches} batches ({time.time() - t0:.1f} s), {ties} answered through the literal heaps, "
      f"{bad} differ from the oracle")
sys.exit(1 if bad else 0)
