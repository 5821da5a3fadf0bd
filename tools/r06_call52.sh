#!/bin/bash
# call 52: one-query kernels with the TRUE next candidate's id row requested before the accept rule (-DHNSW_NEXT_PREFETCH=1, lib_npf.so): parity subset, then A/B
cd "$(dirname "$0")/.."
P=$PWD/hnswlib-rs_amd
HNSW_MI355X_LIB=$P/lib_npf.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -3
CFG=sift1m tools/variant_ab.sh r06_call52 base:10000 npf:10000:LIB=lib_npf.so base2:10000 npf2:10000:LIB=lib_npf.so base100k:100000 npf100k:100000:LIB=lib_npf.so 2>&1 | grep -E "^== |strict qps|^10000 queries|^100000 queries" | cut -c1-200
CFG=glove25 tools/variant_ab.sh r06_call52g base:10000 npf:10000:LIB=lib_npf.so 2>&1 | grep -E "^== |strict qps" | cut -c1-200
