#!/bin/bash
# call 42: the host-buffer boundary again: chunks of the gather / descent (2 is the default since round 4, before the helper threads span by the clock)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call42; mkdir -p $O
timeout 600 python bench.py --config sift1m --steps 2 --warmup 1 --no-boundary --no-cpu-baseline --no-recall --no-traffic --no-concurrent > $O/build.json 2> $O/build.log
HNSWGPU_TRACE_HOST=1 timeout 900 python tools/host_call_sweep.py --config sift1m --reps 60 --build-first 300000 "HNSWGPU_HOST_CHUNKS=1" "HNSWGPU_HOST_CHUNKS=3" "HNSWGPU_HOST_CHUNKS=4" "HNSWGPU_HOST_CHUNKS=6" "HNSWGPU_HOST_CHUNKS=4,HNSWGPU_HOST_THREADS=12" "HNSWGPU_HOST_CHUNKS=3,HNSWGPU_HOST_THREADS=12" 2> $O/sweep.err | tail -30
grep -i "median\|us " $O/sweep.err | tail -12 | cut -c1-300
