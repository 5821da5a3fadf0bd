#!/bin/bash
# Round-3 GPU call 10: replay of the literal candidate heap in runs (default) vs push by push (norun), and what the literal path
# costs per query (per-query durations by status).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call10
mkdir -p $O
for v in default norun default norun; do
  echo "== sift1m $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-boundary --no-concurrent --dump-stats $O/st_sift1m_$v.npy 2>/dev/null | python tools/bench_line.py
done
for v in default norun; do
  echo "== glove25_dot $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 400 python bench.py --config glove25_dot --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary --dump-stats $O/st_dot_$v.npy 2>/dev/null | python tools/bench_line.py
done
python tools/literal_cost.py $O/st_sift1m_default.npy $O/st_sift1m_norun.npy $O/st_dot_default.npy $O/st_dot_norun.npy
