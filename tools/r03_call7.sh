#!/bin/bash
# Round-3 GPU call 7: rows of the speculated next candidate touched one expansion ahead (variant rowpf) against the default build.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for v in default rowpf default rowpf; do
  echo "== sift1m $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-boundary 2>/dev/null | python tools/bench_line.py
done
for cfg in glove25 mnist784; do
  for v in default rowpf; do
    echo "== $cfg $v"
    if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
    timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary 2>/dev/null | python tools/bench_line.py
  done
done
export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_rowpf.so
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
