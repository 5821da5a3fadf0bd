#!/bin/bash
# call 46: where the pair pass starts to pay for strict DistCosine calls (queries per call), then rocprofv3 of config 3 at 100 000 queries per call (default policy)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call46; mkdir -p $O
specs=""
for nq in 15000 20000 30000 40000 60000; do specs="$specs off$nq:$nq:HNSWGPU_PAIR_SEARCH=0 pair$nq:$nq:HNSWGPU_PAIR_SEARCH=1"; done
CFG=glove25 tools/variant_ab.sh r06_call46 $specs 2>&1 | grep -E "^== |strict qps" | cut -c1-200
timeout 900 tools/profile_round.sh r06_call46/prof --config glove25 --nq 100000 > $O/prof.log 2>&1
python tools/summarize_profile.py $O/prof > $O/rocprofv3_summary_glove25_nq100k.txt 2>&1
cp $(find $O/prof/kt -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_glove25_nq100k.csv 2>/dev/null
head -8 $O/rocprofv3_summary_glove25_nq100k.txt
rm -rf $O/prof
