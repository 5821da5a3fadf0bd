#!/bin/bash
# Round-3 GPU call 12: phase timers of the final kernel on configs 2 and 3.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call12
mkdir -p $O
export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_ph.so
for cfg in sift1m glove25; do
  timeout 300 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent --no-boundary --dump-stats $O/ph_$cfg.npy > /dev/null 2>&1
  python tools/phase_report.py $O/ph_$cfg.npy
done
