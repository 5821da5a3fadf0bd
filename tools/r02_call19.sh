#!/bin/bash
# Round-2 GPU call 19: issue priority for searches that turn out long (HNSW_LONG_PRIO = expansions before s_setprio 1, twice that: 2).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call19
mkdir -p $O
for v in default lp64 lp96 default lp64 lp96; do
  echo "== $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2>> $O/bench_$v.log | tee -a $O/bench_sift1m_$v.json | python tools/bench_line.py
done
