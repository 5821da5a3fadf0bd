#!/bin/bash
# Round-3 GPU call 13: re-speculation of the id-row prefetch by a ballot (no wave minimum) against the default; how often the
# prefetch misses.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call13
mkdir -p $O
for v in default respec default respec; do
  echo "== sift1m $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-boundary 2>/dev/null | python tools/bench_line.py
done
for cfg in glove25 glove25_dot mnist784; do
  for v in default respec; do
    echo "== $cfg $v"
    if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
    timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary 2>/dev/null | python tools/bench_line.py
  done
done
for v in ph phrs; do
  export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent --no-boundary --dump-stats $O/$v.npy > /dev/null 2>&1
  python tools/phase_report.py $O/$v.npy | head -1
done
