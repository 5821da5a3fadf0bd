#!/bin/bash
# Round-3 GPU call 19: wave issue priority (s_setprio) for long searches (one level every HNSWGPU_PRIO_STEP expansions) and for
# queries that replay their log (HNSWGPU_PRIO_LITERAL).
# (A record of a measurement: the variants and switches it compares were removed afterwards -- DESIGN.md section 6, "not kept".)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call19
mkdir -p $O
run() {  # cfg step literal extra...
  local cfg=$1 st=$2 lit=$3; shift 3
  export HNSWGPU_PRIO_STEP=$st HNSWGPU_PRIO_LITERAL=$lit
  echo "== $cfg step=$st literal=$lit"
  timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-recall --no-boundary --no-cpu-baseline "$@" 2>/dev/null | python tools/bench_line.py | cut -c1-200
}
run sift1m 0 0 --dump-stats $O/st_sift1m_0_0.npy
run sift1m 0 1 --dump-stats $O/st_sift1m_0_1.npy
run sift1m 96 1 --dump-stats $O/st_sift1m_96_1.npy
run sift1m 64 1 --dump-stats $O/st_sift1m_64_1.npy
run sift1m 128 1 --dump-stats $O/st_sift1m_128_1.npy
run sift1m 48 1
run sift1m 0 0
run glove25_dot 0 0 --no-concurrent
run glove25_dot 0 1 --no-concurrent --dump-stats $O/st_dot_0_1.npy
run glove25_dot 150 1 --no-concurrent
run glove25 0 0 --no-concurrent
run glove25 0 1 --no-concurrent
run glove25 150 1 --no-concurrent
python tools/literal_cost.py $O/st_*.npy
