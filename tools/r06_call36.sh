#!/bin/bash
# call 36: pair kernel: resident workgroups per CU against the round quantisation of a 10 000-query batch; a 100 000-query batch
cd "$(dirname "$0")/.."
CFG=glove25 tools/variant_ab.sh r06_call36 w9:10000:HNSWGPU_PAIR_SEARCH=1,HNSWGPU_PAIR_WG_PER_CU=9 w10:10000:HNSWGPU_PAIR_SEARCH=1,HNSWGPU_PAIR_WG_PER_CU=10 w11:10000:HNSWGPU_PAIR_SEARCH=1,HNSWGPU_PAIR_WG_PER_CU=11 \
   off100k:100000:HNSWGPU_PAIR_SEARCH=0 pair100k:100000:HNSWGPU_PAIR_SEARCH=1 2>&1 | grep -v "^$" | grep -v "last finishers"
CFG=glove25_dot tools/variant_ab.sh r06_call36d w10:10000:HNSWGPU_PAIR_SEARCH=1,HNSWGPU_PAIR_WG_PER_CU=10 2>&1 | grep -v "^$" | grep -v "last finishers"
