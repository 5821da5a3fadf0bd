#!/bin/bash
# call 41: pair kernel with (1) branch-free table tests, (2) cr by votes for <= 4 candidates, (3) the next candidate's id row requested before the accept rule
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call41; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair_search.py -m gpu -x -q 2>&1 | tail -6
CFG=glove25 tools/variant_ab.sh r06_call41 off:10000:HNSWGPU_PAIR_SEARCH=0 pair:10000:HNSWGPU_PAIR_SEARCH=1 ph:10000:HNSWGPU_PAIR_SEARCH=1,LIB=lib_pairph.so off100k:100000:HNSWGPU_PAIR_SEARCH=0 pair100k:100000:HNSWGPU_PAIR_SEARCH=1 2>&1 | grep -v "^$" | grep -v "last finishers" | grep -v "^first round" | cut -c1-330
python tools/pair_phases.py $O/st_ph.npy
CFG=glove25_dot tools/variant_ab.sh r06_call41d pair:10000:HNSWGPU_PAIR_SEARCH=1 pair100k:100000:HNSWGPU_PAIR_SEARCH=1 2>&1 | grep -v "^$" | grep -v "last finishers" | grep -v "^first round" | cut -c1-330
