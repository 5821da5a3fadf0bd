#!/bin/bash
# Round-3 GPU call 20: what the replay of a heap-operation log costs (profiling builds, tools/literal_profile.py): heap operations,
# fences, share of the pops; the literal heap's LDS part at its adaptive size and at 2048 entries.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call20
mkdir -p $O
export HNSWGPU_TRACE_LAUNCH=1
run() {  # cfg variant tag extra...
  local cfg=$1 v=$2 tag=$3; shift 3
  export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so
  echo "== $cfg $v $tag"
  timeout 300 python bench.py --config $cfg --steps 6 --warmup 2 --no-recall --no-boundary --no-cpu-baseline --no-concurrent --dump-stats $O/pf_${cfg}_${v}_$tag.npy 2> $O/err.log | python tools/bench_line.py | cut -c1-200
  grep "hnswgpu launch" $O/err.log | sort | uniq -c | sort -rn | head -3
}
for cfg in sift1m glove25 glove25_dot; do
  run $cfg t2 adaptive
  HNSWGPU_CAND_LDS=2048 run $cfg t2 lds2048
  run $cfg t2old adaptive
done
python tools/literal_profile.py $O/pf_*.npy
