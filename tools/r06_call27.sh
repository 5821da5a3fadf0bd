#!/bin/bash
# call 27: where an expansion of the literal (filtered) kernel spends its time: profiling build -DHNSW_EXACT_PHASES (clock64 around the
# pop, the visited test, the distances, the accept rule, the pushes)
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r06_call27; mkdir -p $O
HNSW_BENCH_DUMP_FILTERED=$O HNSW_MI355X_LIB=$PWD/hnswlib-rs_amd/lib_xph.so timeout 900 python bench.py --steps 2 --warmup 1 --no-recall --no-cpu-baseline --no-traffic --no-concurrent > $O/bench.json 2> $O/err.log
python - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob("gpurun_out/r06_call27/filtered_*.npy")):
    st = np.load(f).astype(np.int64)
    nx = np.maximum(st[:, 1], 1)
    dur = ((st[:, 5] - st[:, 4]) & 0xFFFFFFFF) / 100.0
    ph = {"pop": st[:, 0], "visited": st[:, 2], "distances": st[:, 3], "accept": st[:, 6], "push": st[:, 7]}
    tot = sum(v.sum() for v in ph.values()) * 16
    rate = tot / dur.sum()  # ticks per us, if the phases cover the whole query
    print(f, "queries", len(st), "us/expansion p50 %.2f" % np.median(dur / nx), "ticks covered per us %.0f" % rate)
    for k, v in ph.items():
        print("   %-10s %5.1f %% of the covered ticks, %6.0f ticks per expansion" % (k, 100.0 * v.sum() * 16 / tot, v.sum() * 16 / nx.sum()))
PY
