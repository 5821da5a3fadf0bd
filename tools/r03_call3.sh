#!/bin/bash
# Round-3 GPU call 3: visited probe split into test -> row loads -> insertion (+ packed compare), DistCosine norm inside the row,
# glove25_dot, full-size tests, sharded device entry, 2-rank bench command.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call3
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "multi-rank gather|passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -12
for v in base lazy default inl base lazy default inl; do
  echo "== sift1m $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-boundary 2>/dev/null | python tools/bench_line.py
done
for v in base lazy default inl; do
  echo "== glove25 $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 400 python bench.py --config glove25 --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary 2>/dev/null | python tools/bench_line.py
done
unset HNSW_MI355X_LIB
echo "== glove25_dot default"
timeout 400 python bench.py --config glove25_dot --steps 10 --warmup 3 --no-cpu-baseline --no-concurrent --no-boundary 2>/dev/null | python tools/bench_line.py
for v in base default; do
  echo "== mnist784 $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 400 python bench.py --config mnist784 --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary 2>/dev/null | python tools/bench_line.py
done
echo "== phase timing (sift1m)"
HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_ph.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent --no-boundary --dump-stats $O/ph_stats.npy > /dev/null 2>&1
python tools/phase_report.py $O/ph_stats.npy
echo "== boundary timings (sift1m default)"
unset HNSW_MI355X_LIB
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent 2>$O/boundary.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): print(json.dumps(json.loads(l)['boundary']))"
