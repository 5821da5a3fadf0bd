#!/bin/bash
# Round-3 GPU call 2: lazy literal heap with wave-uniform call results (4 and 5 waves/SIMD) against the round-2 build; GPU suite
# incl. the 2-rank bench command.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call2
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "multi-rank gather|passed|failed|Error" $O/pytest.log | tail -5
for v in base default lb5 base default lb5; do
  echo "== sift1m $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2>/dev/null | python tools/bench_line.py
done
for cfg in glove25 mnist784; do
  for v in base default lb5; do
    echo "== $cfg $v"
    if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
    timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent 2>/dev/null | python tools/bench_line.py
  done
done
