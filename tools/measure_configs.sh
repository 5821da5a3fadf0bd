#!/bin/bash
# Official bench lines of every BASELINE config with the current build, into gpurun_out/<dir>/ (copy what should be
# judged into profiles/).   GPU box: gpurun --timeout 900 -- 'tools/measure_configs.sh r02'
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/${1:-configs}
mkdir -p $O
for cfg in sift1m glove25 glove25_dot mnist784 random10k; do
  timeout 400 python bench.py --config $cfg --steps 20 --warmup 3 > $O/bench_$cfg.json 2> $O/bench_$cfg.log
  echo "== $cfg"; python tools/bench_line.py < $O/bench_$cfg.json
done
