#!/bin/bash
# call 50: pair tests with the final default policy (DistCosine / DistDot, short rows, >= 40 000 queries, strict)
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_pair_search.py -m gpu -x -q 2>&1 | tail -8
HNSWGPU_TRACE_LAUNCH=1 timeout 600 python bench.py --config glove25_dot --nq 100000 --steps 8 --warmup 2 --no-boundary --no-cpu-baseline --no-traffic --no-concurrent > gpurun_out/r06_call50_glove25_dot_nq100k.json 2> gpurun_out/r06_call50.err
python tools/bench_line.py < gpurun_out/r06_call50_glove25_dot_nq100k.json | cut -c1-500
grep "hnswgpu launch" gpurun_out/r06_call50.err | sort | uniq -c | sort -rn | head -2 | cut -c1-160
