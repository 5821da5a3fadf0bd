#!/bin/bash
# call 54: soak at full size on the final tree, default path (strict), configs 2, 3, 3', 5: fresh batches + stored points + filtered queries against the oracle
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call54; mkdir -p $O
for spec in "sift1m 6 6144" "glove25 3 1536" "glove25_dot 3 1536" "mnist784 2 512"; do
  set -- $spec
  timeout 600 python bench.py --config $1 --steps 2 --warmup 1 --no-boundary --no-cpu-baseline --no-recall --no-traffic --no-concurrent > $O/build_$1.json 2> $O/build_$1.log
  timeout 1500 python tools/soak_parity.py --config $1 --batches $2 --points-as-queries 200 --filtered $3 2>&1 | tail -3 | cut -c1-300
done
