#!/bin/bash
# A/B of the opt-in kernel experiments against the default build, in ONE gpurun call (the racy parallel builder gives a
# different graph on every box, so numbers are only comparable within a call).
#   local:   tools/ab_round.sh build            # builds hnswlib-rs_amd/lib_<name>.so for every experiment (CPU, ~2 min each)
#   GPU box: gpurun --timeout 900 -- 'tools/ab_round.sh run'
#   local:   tools/ab_round.sh clean
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
NAMES=(resume valuer endstate specrows cosg)
FLAGS=("-DHNSW_STRICT_RESUME=1" "-DHNSW_EXACT_VALUE_R=1" "-DHNSW_STRICT_RESUME=1 -DHNSW_EXACT_VALUE_R=1" "-DHNSW_SPEC_ROWS=1" "-DHNSW_COSINE_GROUPS=1")
case "${1:-}" in
  build)
    for i in "${!NAMES[@]}"; do tools/mkvariant.sh "${NAMES[$i]}" "${FLAGS[$i]}" || exit 1; done ;;
  clean)
    for n in "${NAMES[@]}"; do rm -f hnswlib-rs_amd/lib_$n.so; done ;;
  run)
    mkdir -p gpurun_out/ab
    echo "== default build: full GPU suite, then the replay-kernel fallback on the tie tests"
    timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
    HNSWGPU_NO_INKERNEL=1 timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "strict_ties or search_matches" 2>&1 | tail -2
    echo "== default build: bench"
    timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --dump-stats gpurun_out/ab/stats_default.npy 2>/dev/null | tee gpurun_out/ab/bench_default.json | python tools/bench_line.py
    for n in "${NAMES[@]}"; do
      lib=$ROOT/hnswlib-rs_amd/lib_$n.so
      [ -f "$lib" ] || continue
      echo "== $n: parity tests, then bench"
      HNSW_MI355X_LIB=$lib timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
      cfg=sift1m; [ "$n" = cosg ] && cfg=glove25
      HNSW_MI355X_LIB=$lib timeout 300 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-recall --dump-stats gpurun_out/ab/stats_$n.npy 2>/dev/null | tee gpurun_out/ab/bench_$n.json | python tools/bench_line.py
    done
    [ -f hnswlib-rs_amd/lib_cosg.so ] && { echo "== default build on glove25 (reference for cosg)"; timeout 300 python bench.py --config glove25 --steps 10 --warmup 2 --no-cpu-baseline --no-recall 2>/dev/null | python tools/bench_line.py; } ;;
  *) echo "usage: $0 build|run|clean"; exit 2 ;;
esac
