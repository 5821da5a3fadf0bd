#!/bin/bash
# Round-2 GPU call 16: merge_list (accept rule for a whole neighbour list at once, S == 1) -- parity suite, then A/B
# against the same build without it (hnswlib-rs_amd/lib_nomerge.so from tools/mkvariant.sh nomerge -DHNSW_MERGE_LISTS=0).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call16
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for v in default nomerge default nomerge; do
  echo "== $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2>> $O/bench_$v.log | tee -a $O/bench_sift1m_$v.json | python tools/bench_line.py
done
for v in default nomerge; do
  echo "== random10k $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config random10k --steps 20 --warmup 4 --no-recall --cpu-seconds 3 2>> $O/bench_r_$v.log | tee -a $O/bench_random10k_$v.json | python tools/bench_line.py
done
