#!/bin/bash
# call 24: what the replay of the heap-operation log costs on HEAD (profiling build -DHNSW_PHASE_TIMING=2, tools/literal_profile.py)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call24; mkdir -p $O
HNSW_MI355X_LIB=$PWD/hnswlib-rs_amd/lib_pt2.so timeout 600 python bench.py --config sift1m --steps 8 --warmup 2 --no-recall --no-boundary --no-cpu-baseline --no-traffic --dump-stats $O/st.npy > $O/bench.json 2> $O/err.log
python tools/literal_profile.py $O/st.npy
