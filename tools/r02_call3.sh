#!/bin/bash
# Round-2 GPU call 3: where does an expansion's time go (phase timers), empty-machine latency, next-round touch A/B.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call3
mkdir -p $O
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $*"; }
bench() {  # name lib config extra...
  local name=$1 lib=$2 cfg=$3; shift 3
  HNSW_MI355X_LIB=$lib timeout 400 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-recall "$@" \
      --dump-stats $O/stats_${name}_$cfg.npy 2> $O/bench_${name}_$cfg.log | tee $O/bench_${name}_$cfg.json | python tools/bench_line.py
}
L=$ROOT/hnswlib-rs_amd
stamp "default sift1m"; bench default "" sift1m
stamp "touch sift1m"; bench touch $L/lib_touch.so sift1m
stamp "default sift1m again"; bench default2 "" sift1m
stamp "phase-timing build, strict"; bench phase $L/lib_phase.so sift1m
stamp "phase-timing + touch"; bench phasetouch $L/lib_phasetouch.so sift1m
python tools/phase_report.py $O/stats_phase_sift1m.npy $O/stats_phasetouch_sift1m.npy
stamp "empty-machine latency: 64 queries per batch"
bench phase64 $L/lib_phase.so sift1m --nq 64
bench phasetouch64 $L/lib_phasetouch.so sift1m --nq 64
python tools/phase_report.py $O/stats_phase64_sift1m.npy $O/stats_phasetouch64_sift1m.npy
stamp "mnist784: default / touch"
bench default "" mnist784
bench touch $L/lib_touch.so mnist784
stamp "glove25: default / touch"
bench default "" glove25
bench touch $L/lib_touch.so glove25
stamp "parity of the touch build (distance sweep, search tests)"
HNSW_MI355X_LIB=$L/lib_touch.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "sweep or search_matches or edge" 2>&1 | tail -3
stamp done
