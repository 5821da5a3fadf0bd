#!/bin/bash
# call 48 (as call 47, the cut looked at through position ef of the mirror): pair kernel with the three tie conditions of hnsw_search_kernel instead of "any equal distance": tests, soak through the pass, configs 3 / 3' with and without it
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call48; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair_search.py -m gpu -x -q 2>&1 | tail -6
for cfg in glove25 glove25_dot sift1m; do
  timeout 600 python bench.py --config $cfg --steps 2 --warmup 1 --no-boundary --no-cpu-baseline --no-recall --no-traffic --no-concurrent > $O/build_$cfg.json 2> $O/build_$cfg.log
  HNSWGPU_PAIR_SEARCH=1 timeout 900 python tools/soak_parity.py --config $cfg --batches 4 --points-as-queries 200 2>&1 | tail -2 | cut -c1-300
done
for cfg in glove25 glove25_dot; do
CFG=$cfg tools/variant_ab.sh r06_call48_$cfg off:10000:HNSWGPU_PAIR_SEARCH=0 pair:10000:HNSWGPU_PAIR_SEARCH=1,HNSWGPU_TRACE_LAUNCH=1 off100k:100000:HNSWGPU_PAIR_SEARCH=0 pair100k:100000:HNSWGPU_PAIR_SEARCH=1,HNSWGPU_TRACE_LAUNCH=1 2>&1 | grep -E "^== |strict qps" | cut -c1-220
grep "hnswgpu launch" gpurun_out/r06_call48_$cfg/err_pair.log | sort | uniq -c | sort -rn | sed -n 2,3p | cut -c1-120
grep "hnswgpu launch" gpurun_out/r06_call48_$cfg/err_pair100k.log | sort | uniq -c | sort -rn | sed -n 2,3p | cut -c1-120
done
