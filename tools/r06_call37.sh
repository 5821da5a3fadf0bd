#!/bin/bash
# call 37: pair kernel on config 2 / config 4's batch (d = 128, ef = 64) and config 3' at 100 000 queries per call
cd "$(dirname "$0")/.."
CFG=sift1m tools/variant_ab.sh r06_call37 off10k:10000:HNSWGPU_PAIR_SEARCH=0 pair10k:10000:HNSWGPU_PAIR_SEARCH=1,HNSWGPU_TRACE_LAUNCH=1 off100k:100000:HNSWGPU_PAIR_SEARCH=0 pair100k:100000:HNSWGPU_PAIR_SEARCH=1 2>&1 | grep -v "^$" | grep -v "last finishers" | cut -c1-400
grep "hnswgpu launch" gpurun_out/r06_call37/err_pair10k.log | sort | uniq -c | sort -rn | head -3 | cut -c1-200
CFG=glove25_dot tools/variant_ab.sh r06_call37d off100k:100000:HNSWGPU_PAIR_SEARCH=0 pair100k:100000:HNSWGPU_PAIR_SEARCH=1 2>&1 | grep -v "^$" | grep -v "last finishers" | cut -c1-400
