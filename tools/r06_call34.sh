#!/bin/bash
# call 34: pair kernel: table size / occupancy knobs on config 3, then its instruction counters (rocprofv3 --pmc passes)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call34; mkdir -p $O
for v in d1:HNSWGPU_PAIR_TBITS_DELTA=1 d1w6:HNSWGPU_PAIR_TBITS_DELTA=1,HNSWGPU_PAIR_WG_PER_CU=6 d0w8:HNSWGPU_PAIR_WG_PER_CU=8 d0w4:HNSWGPU_PAIR_WG_PER_CU=4; do
  tag=${v%%:*}; e=$(echo ${v#*:} | tr ',' ' ')
  env HNSWGPU_PAIR_SEARCH=1 HNSWGPU_TRACE_LAUNCH=1 $e timeout 600 python bench.py --config glove25 --steps 12 --warmup 3 --no-boundary --no-cpu-baseline --no-traffic --no-concurrent --no-recall > $O/glove25_$tag.json 2> $O/glove25_$tag.err
  echo "== glove25 $tag"; python tools/bench_line.py < $O/glove25_$tag.json | cut -c1-200
  grep "hnswgpu launch" $O/glove25_$tag.err | sort | uniq -c | sort -rn | head -3 | cut -c1-200
done
HNSWGPU_PAIR_SEARCH=1 HNSWGPU_PAIR_TBITS_DELTA=1 timeout 900 tools/profile_round.sh r06_call34/prof_pair --config glove25 > $O/prof.log 2>&1
python tools/summarize_profile.py $O/prof_pair > $O/rocprofv3_summary_pair.txt 2>&1
grep -i "pair\|search_kernel<1, 2, 0, true>" $O/rocprofv3_summary_pair.txt | head -60
rm -rf $O/prof_pair
