#!/bin/bash
# call 35: pair kernel with the in-launch bitmap fallback: parity tests, config 3 on / off / four waves per SIMD, phase cycles
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call35; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair_search.py -m gpu -x -q 2>&1 | tail -8
CFG=glove25 tools/variant_ab.sh r06_call35 off:10000:HNSWGPU_PAIR_SEARCH=0 pair:10000:HNSWGPU_PAIR_SEARCH=1,HNSWGPU_TRACE_LAUNCH=1 lb4:10000:HNSWGPU_PAIR_SEARCH=1,LIB=lib_pairlb4.so,HNSWGPU_TRACE_LAUNCH=1 ph:10000:HNSWGPU_PAIR_SEARCH=1,LIB=lib_pairph.so 2>&1 | grep -v "^$"
grep "hnswgpu launch" $O/err_pair.log | sort | uniq -c | sort -rn | head -3 | cut -c1-200
grep "hnswgpu launch" $O/err_lb4.log | sort | uniq -c | sort -rn | head -3 | cut -c1-200
python tools/pair_phases.py $O/st_ph.npy
CFG=glove25_dot tools/variant_ab.sh r06_call35d off:10000:HNSWGPU_PAIR_SEARCH=0 pair:10000:HNSWGPU_PAIR_SEARCH=1 2>&1 | grep -v "^$"
