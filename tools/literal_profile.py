#!/usr/bin/env python3
"""What the replay of a heap-operation log costs, from a profiling build (tools/mkvariant.sh NAME "-DHNSW_PHASE_TIMING=2") run
through `bench.py --dump-stats`.  In such a build stats = [fences executed inside the literal pops, n_expand, ticks inside the
literal pops, status, t_start, t_end, heap operations performed there, ticks inside the replayed pops]; ticks of clock64
(s_memtime).  The tick rate is estimated per query as (ticks in literal pops) / (duration - expansions x the value-only
queries' median time per expansion)."""
import sys

import numpy as np

for path in sys.argv[1:]:
    st = np.load(path).astype(np.int64)
    nexp = np.maximum(st[:, 1], 1)
    dur = ((st[:, 5] - st[:, 4]) & 0xFFFFFFFF) * 1e-2  # us
    fences, ticks, ops, tpops = st[:, 0] & 0xFFFFFFFF, st[:, 2] & 0xFFFFFFFF, st[:, 6] & 0xFFFFFFFF, st[:, 7] & 0xFFFFFFFF
    lit = ops > 0
    print(f"{path}: {len(st)} queries, {lit.sum()} replayed their log; launch span {dur.max():.0f} us (longest query)")
    if lit.sum() == 0:
        continue
    per_exp = np.median(dur[~lit] / nexp[~lit])
    extra = dur[lit] - nexp[lit] * per_exp
    rate = np.median(ticks[lit] / np.maximum(extra, 1.0))
    print(f"  value-only {per_exp:.2f} us per expansion; a replaying query takes {np.median(extra):.0f} us longer (p50; p90 {np.percentile(extra, 90):.0f}) than its expansions"
          f" alone; ticks per us ~ {rate:.0f}")
    print(f"  heap operations per replaying query p50 {np.median(ops[lit]):.0f} p90 {np.percentile(ops[lit], 90):.0f}; ticks per operation p50 {np.median(ticks[lit] / ops[lit]):.0f}"
          f" (= {np.median(ticks[lit] / ops[lit]) / rate:.2f} us); share of the ticks spent in replayed pops {tpops[lit].sum() / ticks[lit].sum():.0%};"
          f" fences per query p50 {np.median(fences[lit]):.0f} ({fences[lit].sum() / ops[lit].sum():.2f} per operation)")
    for i in np.argsort(dur)[-4:][::-1]:
        print(f"    longest: {dur[i]:.0f} us, {nexp[i]} expansions, {ops[i]} heap operations, {ticks[i]} ticks ({tpops[i]} in pops), {fences[i]} fences, status {st[i, 3]}")
