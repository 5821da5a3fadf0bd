#!/bin/bash
# Round-2 GPU call 15: resident waves per CU swept (HNSWGPU_WAVES_PER_CU caps the persistent grid): the batch time is
# bounded below by its longest query, which runs faster on a less crowded CU.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call15
mkdir -p $O
for w in 0 18 16 14 12 10 8; do
  echo "== waves per CU cap $w"
  if [ $w -eq 0 ]; then unset HNSWGPU_WAVES_PER_CU; else export HNSWGPU_WAVES_PER_CU=$w; fi
  timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2> $O/bench_$w.log | tee $O/bench_sift1m_$w.json | python tools/bench_line.py
done
unset HNSWGPU_WAVES_PER_CU
for w in 0 12 8; do
  echo "== glove25, cap $w"
  if [ $w -eq 0 ]; then unset HNSWGPU_WAVES_PER_CU; else export HNSWGPU_WAVES_PER_CU=$w; fi
  timeout 300 python bench.py --config glove25 --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2> $O/bench_g$w.log | tee $O/bench_glove25_$w.json | python tools/bench_line.py
done
