#!/bin/bash
# A/B sweep of the tuning knobs that exist without touching the source, on ONE box (graphs differ per box, so only numbers of one
# call compare): resident workgroups per CU, the whole-list accept threshold, the strict kernel at 5 waves per SIMD with a literal
# heap small enough for 20 workgroups per CU.  Build the variants first (CPU, minutes each):
#   ONLY_METRICS="0" tools/mkvariant.sh ml2  "-DHNSW_MERGE_LISTS=2"
#   ONLY_METRICS="0" tools/mkvariant.sh ml3  "-DHNSW_MERGE_LISTS=3"
#   ONLY_METRICS="0" tools/mkvariant.sh lb5  "-DHNSW_LB_WAVES_STRICT=5"
# then: gpurun --timeout 400 -- tools/probe_knobs.sh [config]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
CFG=${1:-sift1m}
O=gpurun_out/probe_knobs
mkdir -p $O
export HNSWGPU_TRACE_LAUNCH=1
run() {  # tag [ENV=VALUE ...]: runs with the product library unless HNSW_MI355X_LIB is among the assignments
  local tag=$1; shift
  echo "== $CFG $tag"
  env "$@" timeout 300 python bench.py --config $CFG --steps 10 --warmup 3 --no-recall --no-boundary --no-cpu-baseline \
      --dump-stats $O/st_${CFG}_$tag.npy 2> $O/err_$tag.log | python tools/bench_line.py | cut -c1-200
  grep "hnswgpu launch" $O/err_$tag.log | sort | uniq -c | sort -rn | head -2
}
run base X=1
for w in 12 14; do run waves$w HNSWGPU_WAVES_PER_CU=$w; done
for v in ml2 ml3; do [ -f hnswlib-rs_amd/lib_$v.so ] && run $v HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; done
if [ -f hnswlib-rs_amd/lib_lb5.so ]; then
  run lb5 HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_lb5.so
  run lb5_lds192 HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_lb5.so HNSWGPU_CAND_LDS=192
fi
run base_again X=1
python tools/literal_cost.py $O/st_${CFG}_*.npy
