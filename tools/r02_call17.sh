#!/bin/bash
# Round-2 GPU call 17: merge_list only for lists with >= T qualifying neighbours (T = 1 product, 2, 3, 4 variants).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call17
mkdir -p $O
for v in default m2 m3 m4 default m2 m3 m4; do
  echo "== $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2>> $O/bench_$v.log | tee -a $O/bench_sift1m_$v.json | python tools/bench_line.py
done
