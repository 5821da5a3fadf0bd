#!/bin/bash
# call 38: config 3 / 3' strict at five waves per SIMD (20 workgroups per CU: 10 000 equal-length searches in two lock-step rounds instead of three)
cd "$(dirname "$0")/.."
for cfg in glove25 glove25_dot; do
CFG=$cfg tools/variant_ab.sh r06_call38_$cfg base:10000:HNSWGPU_TRACE_LAUNCH=1 lb5w20:10000:LIB=lib_lb5.so,HNSWGPU_STRICT_WG_PER_CU=20,HNSWGPU_TRACE_LAUNCH=1 lb5ndw20:10000:LIB=lib_lb5nd.so,HNSWGPU_STRICT_WG_PER_CU=20,HNSWGPU_TRACE_LAUNCH=1 lb5ndw18:10000:LIB=lib_lb5nd.so,HNSWGPU_STRICT_WG_PER_CU=18 lb5ndw20c256:10000:LIB=lib_lb5nd.so,HNSWGPU_STRICT_WG_PER_CU=20,HNSWGPU_CAND_LDS=256 2>&1 | grep -v "^$" | grep -v "last finishers" | grep -v "^first round" | cut -c1-330
for t in base lb5w20 lb5ndw20; do grep "hnswgpu launch" gpurun_out/r06_call38_$cfg/err_$t.log | sort | uniq -c | sort -rn | head -1 | cut -c1-220; done
done
