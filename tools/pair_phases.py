#!/usr/bin/env python3
"""Phase cycles of hnsw_search_pair_kernel built with -DHNSW_PAIR_PHASES=1 (stats words 4-7 of a bench.py --dump-stats file):
    tools/pair_phases.py st.npy      -> cycles per expansion LOOP ITERATION of a wavefront (two queries), by phase"""
import sys
import numpy as np
st = np.load(sys.argv[1]).reshape(-1, 8).astype(np.int64)
ok = st[:, 3] == 0
exp = st[ok, 1] - ((st[ok, 7] >> 8) & 0xFF) * 0  # (word 7 is overwritten by the profiling build: the descent's lists stay in the count)
ph = st[ok, 4:8]
names = ["selection + id row wait + prefetch", "visited tests + compaction", "row loads + insertions + chains", "accept rule + array rebuilt"]
tot = ph.sum(1)
print(f"{ok.sum()} queries answered by the pair pass, mean lists scanned {exp.mean():.1f}")
for i, n in enumerate(names):
    print(f"  {n:40s} {ph[:, i].sum() / exp.sum():8.0f} cycles per iteration  ({100.0 * ph[:, i].sum() / tot.sum():4.1f} %)")
print(f"  {'all four':40s} {tot.sum() / exp.sum():8.0f}")
