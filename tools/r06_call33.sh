#!/bin/bash
# call 33: hnsw_search_pair_kernel (two queries per wavefront as the first pass of a batch): parity tests, then config 3 / 3' with and without it
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call33; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair_search.py -m gpu -x -q 2>&1 | tail -15
for cfg in glove25 glove25_dot; do
  for v in off:HNSWGPU_PAIR_SEARCH=0 pair:HNSWGPU_PAIR_SEARCH=1; do
    tag=${v%%:*}; e=${v#*:}
    env HNSWGPU_TRACE_LAUNCH=1 $e timeout 600 python bench.py --config $cfg --steps 12 --warmup 3 --no-boundary --no-cpu-baseline --no-traffic --no-concurrent > $O/${cfg}_$tag.json 2> $O/${cfg}_$tag.err
    echo "== $cfg $tag"; python tools/bench_line.py < $O/${cfg}_$tag.json | cut -c1-330
    grep "hnswgpu launch" $O/${cfg}_$tag.err | sort | uniq -c | sort -rn | head -4 | cut -c1-220
  done
done
