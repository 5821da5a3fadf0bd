#!/usr/bin/env python3
"""How a launch of the search kernel ends, query by query, from `bench.py --dump-stats st.npy` (per query: n_dist, n_expand, n_ids,
status, t_start, t_end in 10 ns ticks, ...): workgroups resident over time, the cost of an expansion for queries that ran on the full
machine and on the draining one, the last finishers (long searches of the first round, long searches that started in the second,
searches that replayed their log: status 3), and what a perfect longest-first order would have bought.
    tools/tail_report.py gpurun_out/<tag>/st_base.npy [slots, default 4096]"""
import heapq
import sys

import numpy as np

st = np.load(sys.argv[1]).astype(np.int64)
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_exp, status = np.maximum(st[:, 1], 1), st[:, 3]
t0, t1 = st[:, 4] & 0xFFFFFFFF, st[:, 5] & 0xFFFFFFFF
T0 = t0.min()
start, end = ((t0 - T0) & 0xFFFFFFFF) * 0.01, ((t1 - T0) & 0xFFFFFFFF) * 0.01
dur = end - start
print(f"{len(st)} queries, launch {end.max():.0f} us; expansions per query p50 {np.percentile(n_exp, 50):.0f} p90 {np.percentile(n_exp, 90):.0f} "
      f"p99 {np.percentile(n_exp, 99):.0f} max {n_exp.max()}; us per expansion p10 {np.percentile(dur / n_exp, 10):.2f} p50 {np.percentile(dur / n_exp, 50):.2f} "
      f"p90 {np.percentile(dur / n_exp, 90):.2f}")
first = start < 5
late = start > np.percentile(start, 90)
print(f"first round ({first.sum()} queries): {np.median(dur[first] / n_exp[first]):.2f} us per expansion, expansions p50 {np.percentile(n_exp[first], 50):.0f} "
      f"p99 {np.percentile(n_exp[first], 99):.0f}; the last tenth to start ({late.sum()}): {np.median(dur[late] / n_exp[late]):.2f} us per expansion")
ts = np.arange(0, end.max() + 50, 50)
print("queries in flight every 50 us:", [int(((start <= t) & (end > t)).sum()) for t in ts])
print("last finishers:")
for i in np.argsort(-end)[:12]:
    kind = "replayed its log" if status[i] == 3 else ("first round" if start[i] < 5 else "started later")
    print(f"  query {i:5d}: start {start[i]:5.0f} end {end[i]:5.0f} us, {n_exp[i]:3d} expansions at {dur[i] / n_exp[i]:.2f} us  ({kind})")


def simulate(order, lat):
    h = [0.0] * slots
    heapq.heapify(h)
    last = 0.0
    for i in order:
        t = heapq.heappop(h)
        e = t + n_exp[i] * lat
        last = max(last, e)
        heapq.heappush(h, e)
    return last


lat = float(np.median(dur[first] / n_exp[first]))
actual = np.argsort(start, kind="stable")
print(f"list scheduling on {slots} slots at {lat:.2f} us per expansion: the order that ran {simulate(actual, lat):.0f} us, perfect longest-first "
      f"{simulate(np.argsort(-n_exp), lat):.0f} us, the longest search alone {n_exp.max() * lat:.0f} us, all the work / slots "
      f"{n_exp.sum() * lat / slots:.0f} us")
