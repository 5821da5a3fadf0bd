#!/usr/bin/env python3
"""Who bounds a batch: per-query start / end times of one launch, and for the queries that were handed over to the
literal candidate heap when that happened and how long its rebuild took.  Input: `bench.py --dump-stats x.npy` run with
a timeline build of the library (tools/mkvariant.sh tl -DHNSW_TIMELINE=1; HNSW_MI355X_LIB=.../lib_tl.so): stats =
[n_dist, n_expand, n_expand at the hand-over (0: none), status, t_start, t_end, t_handover, t_rebuilt] (100 MHz clock)."""
import heapq
import sys

import numpy as np

for path in sys.argv[1:]:
    st = np.load(path).astype(np.int64) & 0xFFFFFFFF
    base = st[:, 4].min()
    rel = lambda a: ((a - base) & 0xFFFFFFFF) * 1e-2
    beg, end = rel(st[:, 4]), rel(st[:, 5])
    dur, nexp = end - beg, np.maximum(st[:, 1], 1)
    lit = st[:, 2] != 0
    first = beg < 5
    print(f"{path}: {len(st)} queries, launch {end.max():.0f} us; sum of query times / 4096 wave slots {dur.sum() / 4096:.0f} us, "
          f"/ 5120 slots {dur.sum() / 5120:.0f} us")
    print(f"  queries that started with the launch: {first.sum()}; per expansion {np.median(dur[first & ~lit] / nexp[first & ~lit]):.2f} us "
          f"(median, sorted-array path)")
    i = int(np.argmax(np.where(lit, 0, dur)))
    print(f"  longest sorted-array query: {dur[i]:.0f} us, {nexp[i]} expansions, started at {beg[i]:.0f} us")
    h = [0.0] * 4096
    for d in sorted(dur, reverse=True):
        heapq.heappush(h, heapq.heappop(h) + d)
    print(f"  longest-first schedule of these durations on 4096 slots: {max(h):.0f} us (the launch cannot beat its longest query: {dur.max():.0f} us)")
    print(f"  handed over to the literal candidate heap: {lit.sum()}")
    print("    start   hand-over at  (expansions)  rebuild us   literal expansions  us each | before, us each | total us")
    for i in np.argsort(-dur * lit)[:min(12, int(lit.sum()))]:
        th, tr = rel(st[i:i + 1, 6])[0], rel(st[i:i + 1, 7])[0]
        nb = int(st[i, 2]); na = max(int(nexp[i]) - nb, 1)
        print(f"   {beg[i]:6.0f}   {th - beg[i]:8.0f}   {nb:6d}   {tr - th:12.1f}   {na:10d}   {(end[i] - tr) / na:14.2f} | {(th - beg[i]) / max(nb, 1):8.2f} | {dur[i]:8.0f}")
