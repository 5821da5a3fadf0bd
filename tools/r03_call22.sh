#!/bin/bash
# Round-3 GPU call 22: soak of the end-of-round tree at full size with fresh seeds; 500 queries of every batch are stored points
# (distance 0 to themselves, duplicates' ties).  Every answer compared with the oracle.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call22
mkdir -p $O
for cfg in sift1m glove25_dot mnist784; do
  timeout 200 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-recall --no-boundary --no-concurrent > /dev/null 2> $O/bench_$cfg.log || tail -3 $O/bench_$cfg.log
  case $cfg in sift1m) B=6;; glove25_dot) B=4;; *) B=3;; esac
  timeout 300 python tools/soak_parity.py --config $cfg --batches $B --seed-base 0x51DE0000 --points-as-queries 500 2>&1 | tail -2
done
