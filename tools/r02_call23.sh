#!/bin/bash
# Round-2 GPU call 23: merge_list for S = 4 in strict kernels (config 5) and for S = 2 in lean kernels (config 3, ties flagged).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for v in default s4 default s4; do
  echo "== mnist784 $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config mnist784 --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent 2>/dev/null | python tools/bench_line.py
done
for v in default lean2 default lean2; do
  echo "== glove25 $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config glove25 --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-concurrent 2>/dev/null | python tools/bench_line.py
done
