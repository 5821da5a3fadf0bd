#!/bin/bash
# call 31: long rows with the next round's loads in front of this round's chain (group_dist PIPE, strict kernel with 4 result slots per lane)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call31; mkdir -p $O
P=$PWD/hnswlib-rs_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q -k "mnist or 784 or full_size" 2>&1 | tail -3
for cfg in mnist784_hbm mnist784; do
  for v in pipe:X=1 nopipe:HNSW_MI355X_LIB=$P/lib_nopipe.so pipe2:X=1 nopipe2:HNSW_MI355X_LIB=$P/lib_nopipe.so; do
    tag=${v%%:*}; e=${v#*:}
    env HNSWGPU_TRACE_LAUNCH=1 $e timeout 600 python bench.py --config $cfg --steps 12 --warmup 3 --no-boundary --no-cpu-baseline --no-traffic --no-concurrent > $O/${cfg}_$tag.json 2> $O/${cfg}_$tag.err
    echo "== $cfg $tag"; python tools/bench_line.py < $O/${cfg}_$tag.json | cut -c1-260
    grep "hnswgpu launch" $O/${cfg}_$tag.err | sort | uniq -c | sort -rn | head -1 | cut -c1-220
  done
done
