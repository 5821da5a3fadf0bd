#!/bin/bash
# Round-2 GPU call 4: issue-priority knobs (runtime, env) on the strict kernel.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call4
mkdir -p $O
run() {  # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall --dump-stats $O/stats_$label.npy 2> $O/bench_$label.log | tee $O/bench_$label.json | python tools/bench_line.py | sed "s/^/$label: /"
}
run base X=1
run after60 HNSWGPU_PRIO_AFTER=60
run after100 HNSWGPU_PRIO_AFTER=100
run after140 HNSWGPU_PRIO_AFTER=140
run first500 HNSWGPU_PRIO_FIRST=500
run first2000 HNSWGPU_PRIO_FIRST=2000
run first2000_after100 HNSWGPU_PRIO_FIRST=2000 HNSWGPU_PRIO_AFTER=100
run first4096_after120 HNSWGPU_PRIO_FIRST=4096 HNSWGPU_PRIO_AFTER=120
run base2 X=1
