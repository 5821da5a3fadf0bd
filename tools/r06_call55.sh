#!/bin/bash
# call 55: config 3 / 3' strict at 20 workgroups per CU (five waves per SIMD, 128-entry heap top: 7 KB of LDS): 5 120 slots = two lock-step rounds for 10 000 queries
cd "$(dirname "$0")/.."
for cfg in glove25 glove25_dot; do
CFG=$cfg tools/variant_ab.sh r06_call55_$cfg base:10000 c128:10000:LIB=lib_lb5nd.so,HNSWGPU_STRICT_WG_PER_CU=20,HNSWGPU_CAND_LDS=128,HNSWGPU_TRACE_LAUNCH=1 c64:10000:LIB=lib_lb5nd.so,HNSWGPU_STRICT_WG_PER_CU=20,HNSWGPU_CAND_LDS=64 c128w19:10000:LIB=lib_lb5nd.so,HNSWGPU_STRICT_WG_PER_CU=19,HNSWGPU_CAND_LDS=128 2>&1 | grep -E "^== |strict qps|queries in flight" | cut -c1-260
grep "hnswgpu launch" gpurun_out/r06_call55_$cfg/err_c128.log | sort | uniq -c | sort -rn | head -1 | cut -c1-200
done
