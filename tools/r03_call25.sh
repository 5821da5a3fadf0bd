#!/bin/bash
# Round-3 GPU call 25: whole-list accept threshold (HNSW_MERGE_LISTS 1 = product, 2, 3) and the batch scheduling switched off.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call25
mkdir -p $O
run() {
  local tag=$1; shift
  echo "== sift1m $tag"
  env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-recall --no-boundary --no-cpu-baseline --no-concurrent 2>/dev/null | python tools/bench_line.py | cut -c1-160
}
run base X=1
run ml2 HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_ml2.so
run ml3 HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_ml3.so
run nosched HNSWGPU_NO_SCHED=1
run base_again X=1
