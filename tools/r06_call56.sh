#!/bin/bash
# call 56: pair kernel with the per-half scalars picked by v_bfe_u32 from a packed register (half_pick): tests, soak through the pass, configs 3 / 3' at 100 000 and 10 000
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call56; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair_search.py -m gpu -x -q 2>&1 | tail -4
for cfg in glove25 glove25_dot; do
  timeout 600 python bench.py --config $cfg --steps 2 --warmup 1 --no-boundary --no-cpu-baseline --no-recall --no-traffic --no-concurrent > $O/build_$cfg.json 2> $O/build_$cfg.log
  HNSWGPU_PAIR_SEARCH=1 timeout 900 python tools/soak_parity.py --config $cfg --batches 3 --points-as-queries 200 2>&1 | tail -1 | cut -c1-200
  CFG=$cfg tools/variant_ab.sh r06_call56_$cfg pair:10000:HNSWGPU_PAIR_SEARCH=1 pair100k:100000 pair100kb:100000 2>&1 | grep -E "^== |strict qps" | cut -c1-200
done
