#!/bin/bash
# Round-2 GPU call 5: bucketed visited table -- parity, bench, phase timers.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call5
mkdir -p $O
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $*"; }
bench() {  # name lib config extra...
  local name=$1 lib=$2 cfg=$3; shift 3
  HNSW_MI355X_LIB=$lib timeout 400 python bench.py --config $cfg --steps 20 --warmup 4 --no-cpu-baseline --no-recall "$@" \
      --dump-stats $O/stats_${name}_$cfg.npy 2> $O/bench_${name}_$cfg.log | tee $O/bench_${name}_$cfg.json | python tools/bench_line.py
}
stamp "GPU suite"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
L=$ROOT/hnswlib-rs_amd
stamp "default sift1m"; bench default "" sift1m
stamp "phase-timing build"; bench phase $L/lib_phase.so sift1m
python tools/phase_report.py $O/stats_phase_sift1m.npy
bench phase64 $L/lib_phase.so sift1m --nq 64
python tools/phase_report.py $O/stats_phase64_sift1m.npy
stamp "mnist784 / glove25"
bench default "" mnist784
bench default "" glove25
stamp done
