#!/bin/bash
# Round-3 GPU call 18: literal candidate heap -- stores to its global part fenced only before a load from it, the id row of the
# entry about to be popped requested before the heap work (default) against the previous form (lib_old.so), the fence alone
# (lib_fence.so); what a literal pop costs (profiling builds t_old / t_new).
# (A record of a measurement: the variants and switches it compares were removed afterwards -- DESIGN.md section 6, "not kept".)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call18
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "exact_first or strict_ties" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5
run() {  # cfg variant extra...
  local cfg=$1 v=$2; shift 2
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  echo "== $cfg $v"
  timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-recall --no-boundary --no-concurrent "$@" 2>/dev/null | python tools/bench_line.py | cut -c1-330
}
for cfg in sift1m glove25_dot; do
  run $cfg old --no-cpu-baseline
  run $cfg default --cpu-seconds 2 --dump-stats $O/st_${cfg}_default.npy
  run $cfg fence --no-cpu-baseline
  run $cfg old --no-cpu-baseline
  run $cfg default --no-cpu-baseline
  run $cfg t_old --no-cpu-baseline --dump-stats $O/pf_${cfg}_old.npy
  run $cfg t_new --no-cpu-baseline --dump-stats $O/pf_${cfg}_new.npy
done
python tools/literal_profile.py $O/pf_*.npy
python tools/literal_cost.py $O/st_*.npy
