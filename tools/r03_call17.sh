#!/bin/bash
# Round-3 GPU call 17: the previous expansion's heap operations handed to the literal pop in registers (default), the literal
# candidate heap kept once it exists (HNSWGPU_EXACT_FIRST=2), and the strict kernel bounded to 5 waves per SIMD (lib_lb5.so).
# (A record of a measurement: the variants and switches it compares were removed afterwards -- DESIGN.md section 6, "not kept".)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call17
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q -x -k "exact_first or strict_ties or config2" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5
run() {  # cfg variant sticky extra...
  local cfg=$1 v=$2 st=$3; shift 3
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  if [ $st = 0 ]; then unset HNSWGPU_EXACT_FIRST; else export HNSWGPU_EXACT_FIRST=$st; fi
  echo "== $cfg $v sticky=$st"
  timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-recall --no-boundary --no-concurrent "$@" 2>/dev/null | python tools/bench_line.py | cut -c1-330
}
for cfg in sift1m glove25_dot; do
  run $cfg default 0 --no-cpu-baseline --dump-stats $O/st_${cfg}_default_0.npy
  run $cfg default 2 --no-cpu-baseline --dump-stats $O/st_${cfg}_default_2.npy
  run $cfg lb5 0 --no-cpu-baseline --dump-stats $O/st_${cfg}_lb5_0.npy
  run $cfg lb5 2 --cpu-seconds 2 --dump-stats $O/st_${cfg}_lb5_2.npy
done
run glove25 default 0 --no-cpu-baseline
run glove25 lb5 2 --cpu-seconds 2
python tools/literal_cost.py $O/st_*.npy
