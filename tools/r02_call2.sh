#!/bin/bash
# Round-2 GPU call 2: the rewritten strict kernel (deferred hand-over), new entry points, full GPU suite.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call2
mkdir -p $O
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $*"; }
stamp "GPU suite"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
stamp "replay-kernel fallback on the tie tests"
HNSWGPU_NO_INKERNEL=1 timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "strict_ties or search_matches" 2>&1 | tail -3
bench() {  # name lib config extra...
  local name=$1 lib=$2 cfg=$3; shift 3
  HNSW_MI355X_LIB=$lib timeout 400 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-recall "$@" \
      --dump-stats $O/stats_${name}_$cfg.npy 2> $O/bench_${name}_$cfg.log | tee $O/bench_${name}_$cfg.json | python tools/bench_line.py
}
stamp "default: sift1m"
bench default "" sift1m
if [ -f hnswlib-rs_amd/lib_w5.so ]; then
  stamp "w5 (strict kernel at 5 waves/SIMD): sift1m"
  bench w5 $ROOT/hnswlib-rs_amd/lib_w5.so sift1m
fi
stamp "default: sift1m with parity at full size"
timeout 400 python bench.py --config sift1m --steps 10 --warmup 2 --no-recall 2> $O/bench_parity_sift1m.log | tee $O/bench_parity_sift1m.json | python tools/bench_line.py
stamp "default: mnist784, glove25"
bench default "" mnist784
bench default "" glove25
stamp done
