#!/bin/bash
# Round-3 GPU call 1: the lazy literal candidate heap (strict kernel): GPU suite, then config 2 / 3 / 5 against the round-2 build.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== sift1m lazy (with parity vs oracle)"
timeout 600 python bench.py --steps 20 --warmup 4 > $O/bench_sift1m_lazy.json 2> $O/bench_sift1m_lazy.log; python tools/bench_line.py < $O/bench_sift1m_lazy.json
for v in base lb5 default base lb5 default; do
  echo "== sift1m $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2>/dev/null | python tools/bench_line.py
done
for cfg in glove25 mnist784; do
  for v in base lb5 default; do
    echo "== $cfg $v"
    if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
    timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent 2>/dev/null | python tools/bench_line.py
  done
done
