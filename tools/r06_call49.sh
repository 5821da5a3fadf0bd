#!/bin/bash
# call 49: the pair pass's crossover for strict DistDot calls (config 3'), and for DistCosine again with the three tie conditions
cd "$(dirname "$0")/.."
specs=""
for nq in 30000 40000 60000; do specs="$specs off$nq:$nq:HNSWGPU_PAIR_SEARCH=0 pair$nq:$nq:HNSWGPU_PAIR_SEARCH=1"; done
CFG=glove25_dot tools/variant_ab.sh r06_call49d $specs 2>&1 | grep -E "^== |strict qps" | cut -c1-200
CFG=glove25 tools/variant_ab.sh r06_call49 off25000:25000:HNSWGPU_PAIR_SEARCH=0 pair25000:25000:HNSWGPU_PAIR_SEARCH=1 off30000:30000:HNSWGPU_PAIR_SEARCH=0 pair30000:30000:HNSWGPU_PAIR_SEARCH=1 2>&1 | grep -E "^== |strict qps" | cut -c1-200
