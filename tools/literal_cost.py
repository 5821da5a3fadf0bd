#!/usr/bin/env python3
"""What the literal-heap path costs a strict launch: per-query durations of `bench.py --dump-stats` split by how the query
was answered (stats = [n_dist, n_expand, n_ids, status, t_start, t_end (100 MHz ticks), bitmap_used, flags]; status 3 =
answered with the literal heaps, flags bit 1 = a pop came from the literal candidate heap)."""
import sys

import numpy as np

for path in sys.argv[1:]:
    st = np.load(path).astype(np.int64)
    dur = ((st[:, 5] - st[:, 4]) & 0xFFFFFFFF) * 1e-2  # us
    t0 = (st[:, 4] & 0xFFFFFFFF)
    beg = ((t0 - t0.min()) & 0xFFFFFFFF) * 1e-2
    end = beg + dur
    nexp = np.maximum(st[:, 1], 1)
    lit_pop = (st[:, 7] & 2) != 0
    lit_any = st[:, 3] == 3
    print(f"{path}: {len(st)} queries, launch span {end.max():.0f} us; last 5 % of the queries end after {np.percentile(end, 95):.0f} us")
    for name, m in (("by values only", ~lit_any), ("literal return_points only (C)", lit_any & ~lit_pop), ("literal candidate pops (A)/(B)", lit_pop)):
        if m.sum() == 0:
            continue
        print(f"  {name:34s} n={m.sum():5d}  expansions p50 {np.median(nexp[m]):5.0f}  duration p50 {np.median(dur[m]):7.1f} us  p99 {np.percentile(dur[m], 99):7.1f}"
              f"  max {dur[m].max():7.1f}  us/expansion p50 {np.median(dur[m] / nexp[m]):5.2f}  latest end {end[m].max():7.0f} us")
