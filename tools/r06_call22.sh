#!/bin/bash
# call 22: the bucketed work list: tests, then A/B against the sort kernel on one box
# (ran on the tree archived in profiles/r06_bucket_worklist/tree_as_measured.diff: the knobs / variant libraries it names are not in HEAD)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_parity.py -m gpu -x -q -k "not full_size" 2>&1 | tail -12 | tee $O/tests.log
pre() { python - "$1" <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=j["roofline"]; print("   pre-kernels ms", r["descent_and_order_kernels"]["ms"], "kernel_ms", r["kernel_ms"], "all", r["all_kernels_ms"], "ms_per_step", j["ms_per_step"], "value", j["value"])
PY
}
export HNSWGPU_TRACE_LAUNCH=1
tools/variant_ab.sh r06_call22 auto10k:10000 order10k:10000:HNSWGPU_SCHED=order auto10kb:10000 order10kb:10000:HNSWGPU_SCHED=order auto12k:12500 order12k:12500:HNSWGPU_SCHED=order auto100k:100000 order100k:100000:HNSWGPU_SCHED=order > $O/ab.log 2>&1
for n in auto10k order10k auto10kb order10kb auto12k order12k auto100k order100k; do echo $n; pre $O/bench_$n.json; grep "hnswgpu launch" $O/err_$n.log | sed 's/.*work list/work list/' | sort | uniq -c | sort -rn | head -3; done
CFG=glove25 tools/variant_ab.sh r06_call22g gauto:10000 gorder:10000:HNSWGPU_SCHED=order > $O/abg.log 2>&1
for n in gauto gorder; do echo $n; pre gpurun_out/r06_call22g/bench_$n.json; done
