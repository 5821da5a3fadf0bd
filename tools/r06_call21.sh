#!/bin/bash
# call 21: the call's timing events bound to the kernels (hipExtLaunchKernelGGL) against markers between them
# (ran on the tree archived in profiles/r06_bucket_worklist/tree_as_measured.diff: the knobs / variant libraries it names are not in HEAD)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round6.py -m gpu -x -q -k "not full_size" 2>&1 | tail -4 | tee $O/tests.log
pre() { python - "$1" <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=j["roofline"]; print("   pre-kernels ms", r["descent_and_order_kernels"]["ms"], "kernel_ms", r["kernel_ms"], "all", r["all_kernels_ms"], "ms_per_step", j["ms_per_step"], "value", j["value"])
PY
}
tools/variant_ab.sh r06_call21 bound10k:10000 marker10k:10000:HNSWGPU_MARKER_EVENTS=1 bound10kb:10000 marker10kb:10000:HNSWGPU_MARKER_EVENTS=1 bound12k:12500 marker12k:12500:HNSWGPU_MARKER_EVENTS=1 > $O/ab.log 2>&1
for n in bound10k marker10k bound10kb marker10kb bound12k marker12k; do echo $n; pre $O/bench_$n.json; done
