#!/usr/bin/env python3
"""Per-expansion cost of the search kernel's phases from a profiling build (make TUNE=-DHNSW_PHASE_TIMING=1) run through
`bench.py --dump-stats`.  In such a build stats = [expansions whose id row was not prefetched, n_expand, gather_cycles, dist_cycles, t_start, t_end,
insert_cycles, select_cycles] (cycle counts of clock64, summed over the query's expansions)."""
import sys

import numpy as np

for path in sys.argv[1:]:
    st = np.load(path).astype(np.int64)
    nexp = np.maximum(st[:, 1], 1)
    dur_us = ((st[:, 5] - st[:, 4]) & 0xFFFFFFFF) * 1e-2
    sel, dist, gather, ins = st[:, 7] & 0xFFFFFFFF, st[:, 3] & 0xFFFFFFFF, st[:, 2] & 0xFFFFFFFF, st[:, 6] & 0xFFFFFFFF
    tot = sel + gather + ins
    t0 = st[:, 4] & 0xFFFFFFFF
    beg = ((t0 - t0.min()) & 0xFFFFFFFF) * 1e-2
    late = beg > np.percentile(beg, 90)     # queries that ran (partly) on a draining machine
    for name, m in (("all", np.ones(len(st), bool)), ("first-round", beg < 5), ("late starters", late)):
        f = lambda a: np.median(a[m] / nexp[m])
        print(f"{path} [{name}, {m.sum()} queries]: per expansion: {f(dur_us):.2f} us | cycles: select+ids {f(sel):.0f}, "
              f"visited+compaction {f(gather - dist):.0f}, rows+distances {f(dist):.0f}, inserts {f(ins):.0f}, sum {f(tot):.0f}; "
              f"clock/us ~ {np.median(tot[m] / np.maximum(dur_us[m], 1e-3)):.0f}; id-row prefetch missed in {st[m, 0].sum() / nexp[m].sum():.1%} of the expansions")
