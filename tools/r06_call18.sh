#!/bin/bash
# call 18: the descent with two queries per wavefront: tests, then A/B by knob on one box
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "descent" 2>&1 | tail -25 | tee $O/tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 | tee -a $O/tests.log
pre() { python - "$1" <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=j["roofline"]; print("   pre-kernels ms", r.get("descent_and_order_kernels"), "kernel_ms", r["kernel_ms"], "all", r["all_kernels_ms"], "value", j["value"])
PY
}
tools/variant_ab.sh r06_call18 pair10k:10000 single10k:10000:HNSWGPU_NO_PAIR_DESCENT=1 pair10kb:10000 single10kb:10000:HNSWGPU_NO_PAIR_DESCENT=1 pair100k:100000 single100k:100000:HNSWGPU_NO_PAIR_DESCENT=1
for n in pair10k single10k pair10kb single10kb pair100k single100k; do echo $n; pre $O/bench_$n.json; done
CFG=glove25 tools/variant_ab.sh r06_call18g gpair:10000 gsingle:10000:HNSWGPU_NO_PAIR_DESCENT=1
for n in gpair gsingle; do echo $n; pre gpurun_out/r06_call18g/bench_$n.json; done
