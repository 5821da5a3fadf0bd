#!/bin/bash
# call 53 (as call 51): the round's record on the final tree (as `gpu_call.sh record r06`), then configs 3 / 3' at 100 000 queries per call (default policy, all launches priced)
cd "$(dirname "$0")/.."
tools/gpu_call.sh record r06
for cfg in glove25 glove25_dot; do
  timeout 600 python bench.py --config $cfg --nq 100000 --steps 8 --warmup 2 --no-boundary --no-cpu-baseline --no-traffic --no-concurrent > gpurun_out/r06_record/bench_${cfg}_nq100k.json 2> gpurun_out/r06_record/bench_${cfg}_nq100k.log
  python tools/bench_line.py < gpurun_out/r06_record/bench_${cfg}_nq100k.json | cut -c1-300
done
