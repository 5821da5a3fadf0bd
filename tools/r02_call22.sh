#!/bin/bash
# Round-2 GPU call 22: merge_list generalised over result slots, on for S = 2 in strict kernels (config 3's shape) -- parity
# suite, then A/B against the same build with -DHNSW_MERGE_S2=0 (lib_nos2.so).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in default nos2 default nos2; do
  echo "== $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config glove25 --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-concurrent 2>/dev/null | python tools/bench_line.py
  timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-concurrent 2>/dev/null | python tools/bench_line.py
done
