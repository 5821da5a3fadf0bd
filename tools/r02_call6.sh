#!/bin/bash
# Round-2 GPU call 6: bucket displacement fix on all configs; occupancy variants.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call6
mkdir -p $O
bench() {  # name lib config extra...
  local name=$1 lib=$2 cfg=$3; shift 3
  HNSW_MI355X_LIB=$lib timeout 400 python bench.py --config $cfg --steps 20 --warmup 4 --no-cpu-baseline --no-recall "$@" \
      --dump-stats $O/stats_${name}_$cfg.npy 2> $O/bench_${name}_$cfg.log | tee $O/bench_${name}_$cfg.json | python tools/bench_line.py | sed "s/^/$name $cfg: /"
}
L=$ROOT/hnswlib-rs_amd
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "overflow or full_size or search_matches or edge" 2>&1 | tail -3
bench default "" sift1m
bench lean4 $L/lib_lean4.so sift1m
bench w5 $L/lib_w5.so sift1m
bench default "" mnist784
bench default "" glove25
bench default "" random10k
