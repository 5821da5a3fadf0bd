#!/bin/bash
# call 19: where the time in front of the search kernel goes: kernel trace with timestamps, pair / single descent
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r06_call19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary --no-traffic"
rocprofv3 --kernel-trace -d $O/kt_pair --output-format csv -- $B > $O/pair.log 2>&1
HNSWGPU_NO_PAIR_DESCENT=1 rocprofv3 --kernel-trace -d $O/kt_single --output-format csv -- $B > $O/single.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for tag in ("pair", "single"):
    f = glob.glob(f"gpurun_out/r06_call19/kt_{tag}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # sequences descend -> order -> search<..true> with the strict kernel: durations and gaps
    out = collections.defaultdict(list)
    for i in range(len(rows) - 2):
        a, b, c = rows[i], rows[i + 1], rows[i + 2]
        if "descend" in a["Kernel_Name"] and "order_desc" in b["Kernel_Name"] and "hnsw_search_kernel" in c["Kernel_Name"]:
            s = lambda r: int(r["Start_Timestamp"]); e = lambda r: int(r["End_Timestamp"])
            out["descend"].append(e(a) - s(a)); out["gap1"].append(s(b) - e(a)); out["order"].append(e(b) - s(b))
            out["gap2"].append(s(c) - e(b)); out["search"].append(e(c) - s(c)); out["grid_descend"].append(int(a["Grid_Size"]) if "Grid_Size" in a else 0)
    import statistics as st
    print(tag, {k: round(st.median(v) / 1000, 2) for k, v in out.items()}, "n", len(out["descend"]))
PY
rm -rf $O/kt_pair $O/kt_single
