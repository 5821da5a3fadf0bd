#!/bin/bash
# call 20: descent kernel durations from a kernel trace: product, two priority variants, one-query kernel
# (ran on the tree archived in profiles/r06_bucket_worklist/tree_as_measured.diff: the knobs / variant libraries it names are not in HEAD)
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r06_call20; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "descent" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary --no-traffic"
run() { tag=$1; shift; env "$@" rocprofv3 --kernel-trace -d $O/kt_$tag --output-format csv -- $B > $O/$tag.log 2>&1; }
run product X=1
run prio1 HNSW_MI355X_LIB=$R/hnswlib-rs_amd/lib_dprio1.so
run prio2 HNSW_MI355X_LIB=$R/hnswlib-rs_amd/lib_dprio2.so
run single HNSWGPU_NO_PAIR_DESCENT=1
cd $R
python - <<'PY'
import csv, glob, collections, statistics as st
for tag in ("product", "prio1", "prio2", "single"):
    f = glob.glob(f"gpurun_out/r06_call20/kt_{tag}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    out = collections.defaultdict(list)
    for i in range(len(rows) - 2):
        a, b, c = rows[i], rows[i + 1], rows[i + 2]
        if "descend" in a["Kernel_Name"] and "order_desc" in b["Kernel_Name"] and "hnsw_search_kernel" in c["Kernel_Name"]:
            s = lambda r: int(r["Start_Timestamp"]); e = lambda r: int(r["End_Timestamp"])
            out["descend"].append(e(a) - s(a)); out["gap1"].append(s(b) - e(a)); out["order"].append(e(b) - s(b))
            out["gap2"].append(s(c) - e(b)); out["search"].append(e(c) - s(c))
    print(tag, {k: round(st.median(v) / 1000, 2) for k, v in out.items()}, "n", len(out["descend"]))
PY
rm -rf $O/kt_*
