#!/bin/bash
# Round-2 GPU call 7: GPU-assisted construction (tests, then the 1M x 128 build inside bench.py), cand_lds = 512.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call7
mkdir -p $O
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $*"; }
stamp "construction tests"
timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "construction" 2>&1 | tail -15
stamp "sift1m, GPU-assisted build"
timeout 600 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline 2> $O/bench_gpubuild_sift1m.log | tee $O/bench_gpubuild_sift1m.json | python tools/bench_line.py
grep -E "built|building" $O/bench_gpubuild_sift1m.log
python - <<'PY'
import json
j=json.load(open("gpurun_out/r02_call7/bench_gpubuild_sift1m.json")); print("recall", j["recall_at_10"], "setup", j["setup_s"])
PY
stamp "sift1m, host build (same box)"
timeout 600 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --host-build 2> $O/bench_hostbuild_sift1m.log | tee $O/bench_hostbuild_sift1m.json | python tools/bench_line.py
python - <<'PY'
import json
j=json.load(open("gpurun_out/r02_call7/bench_hostbuild_sift1m.json")); print("recall", j["recall_at_10"], "setup", j["setup_s"])
PY
stamp done
