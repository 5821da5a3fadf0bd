#!/bin/bash
# call 43: soak at full size THROUGH the pair pass (HNSWGPU_PAIR_SEARCH=1: configs 3, 3', 2), then the default policy on config 3 at 100 000 queries per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call43; mkdir -p $O
for cfg in glove25 glove25_dot sift1m; do
  timeout 600 python bench.py --config $cfg --steps 2 --warmup 1 --no-boundary --no-cpu-baseline --no-recall --no-traffic --no-concurrent > $O/build_$cfg.json 2> $O/build_$cfg.log
  HNSWGPU_PAIR_SEARCH=1 timeout 900 python tools/soak_parity.py --config $cfg --batches 3 --points-as-queries 200 2>&1 | tail -3 | cut -c1-300
done
HNSWGPU_TRACE_LAUNCH=1 timeout 600 python bench.py --config glove25 --nq 100000 --steps 8 --warmup 2 --no-boundary --no-cpu-baseline --no-traffic --no-concurrent > $O/glove25_nq100k_default.json 2> $O/glove25_nq100k_default.err
python tools/bench_line.py < $O/glove25_nq100k_default.json | cut -c1-600
grep "hnswgpu launch" $O/glove25_nq100k_default.err | sort | uniq -c | sort -rn | head -3 | cut -c1-200
