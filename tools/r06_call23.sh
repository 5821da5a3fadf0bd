#!/bin/bash
# call 23: kernel trace of the bucketed work list against the sort kernel
# (ran on the tree archived in profiles/r06_bucket_worklist/tree_as_measured.diff: the knobs / variant libraries it names are not in HEAD)
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r06_call23; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary --no-traffic"
run() { tag=$1; shift; env "$@" rocprofv3 --kernel-trace -d $O/kt_$tag --output-format csv -- $B > $O/$tag.log 2>&1; }
run auto X=1
run order HNSWGPU_SCHED=order
cd $R
python - <<'PY'
import csv, glob, collections, statistics as st
for tag in ("auto", "order"):
    f = glob.glob(f"gpurun_out/r06_call23/kt_{tag}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    s = lambda r: int(r["Start_Timestamp"]); e = lambda r: int(r["End_Timestamp"])
    out = collections.defaultdict(list)
    for i in range(len(rows) - 2):
        a, b, c = rows[i], rows[i + 1], rows[i + 2]
        if "descend" in a["Kernel_Name"] and "order_desc" in b["Kernel_Name"] and "hnsw_search_kernel" in c["Kernel_Name"] and "true" in c["Kernel_Name"]:
            out["S descend"].append(e(a) - s(a)); out["S order"].append(e(b) - s(b)); out["S gap2"].append(s(c) - e(b)); out["S search"].append(e(c) - s(c))
        if "descend" in a["Kernel_Name"] and "hnsw_search_kernel" in b["Kernel_Name"] and "true" in b["Kernel_Name"]:
            out["B descend"].append(e(a) - s(a)); out["B gap"].append(s(b) - e(a)); out["B search"].append(e(b) - s(b))
    print(tag, {k: (round(st.median(v) / 1000, 2), len(v)) for k, v in out.items()})
PY
rm -rf $O/kt_*
