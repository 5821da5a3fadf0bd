#!/bin/bash
# Round-2 GPU call 21: quiet compute units for literal searches (HNSW_QUIET_CU=48 variant) against the product build.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call21
mkdir -p $O
for v in default q48 default q48; do
  echo "== $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2>> $O/bench_$v.log | tee -a $O/bench_sift1m_$v.json | python tools/bench_line.py
done
for v in default q48; do
  echo "== mnist784, glove25: $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config mnist784 --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent 2>> $O/bench_$v.log | python tools/bench_line.py
  timeout 300 python bench.py --config glove25 --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-concurrent 2>> $O/bench_$v.log | python tools/bench_line.py
done
HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_q48tl.so timeout 300 python bench.py --config sift1m --steps 8 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent --dump-stats $O/stats_q48tl.npy 2>> $O/bench.log | python tools/bench_line.py
python tools/timeline_report.py $O/stats_q48tl.npy | head -14
