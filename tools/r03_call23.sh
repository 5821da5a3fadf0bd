#!/bin/bash
# Round-3 GPU call 23: the soak of call 22 for config 3 (DistCosine, norm inside the row) and config 1.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call23
mkdir -p $O
for cfg in glove25 random10k; do
  timeout 120 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-recall --no-boundary --no-concurrent > /dev/null 2> $O/bench_$cfg.log || tail -3 $O/bench_$cfg.log
  timeout 120 python tools/soak_parity.py --config $cfg --batches 4 --seed-base 0x51DE0000 --points-as-queries 300 2>&1 | tail -2
done
