#!/bin/bash
# Round-2 GPU call 20: merge_list + heap_push_many in the literal continuation (own LDS for the merge buffer) -- parity
# suite, then A/B against the previous build (hnswlib-rs_amd/lib_base.so).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call20
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for v in default base default base; do
  echo "== $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2>> $O/bench_$v.log | tee -a $O/bench_sift1m_$v.json | python tools/bench_line.py
done
for v in default base; do
  echo "== mnist784 $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config mnist784 --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent 2>> $O/bench_$v.log | python tools/bench_line.py
  timeout 300 python bench.py --config glove25 --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-concurrent 2>> $O/bench_$v.log | python tools/bench_line.py
done
