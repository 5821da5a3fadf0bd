#!/bin/bash
# Round-3 GPU call 4: late re-speculation of the id-row prefetch (default) against the same build without it; the fixed tests;
# where the host side of the GPU-assisted build spends its time.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call4
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -s -k "cosine_norm or bench_gpus_2 or sharded_device" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "multi-rank gather|passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -12
for v in nors default nors default; do
  echo "== sift1m $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-boundary 2>/dev/null | python tools/bench_line.py
done
for cfg in glove25 glove25_dot mnist784; do
  for v in nors default; do
    echo "== $cfg $v"
    if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
    timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary 2>/dev/null | python tools/bench_line.py
  done
done
unset HNSW_MI355X_LIB
echo "== build timing (1M x 128, fresh cache)"
HNSWGPU_BUILD_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-concurrent --no-boundary --cache-dir /tmp/fresh_cache 2>&1 | grep -E "hnswgpu build|built in" 
