#!/bin/bash
# round 6, call 7: sixteen row loads in flight for lists of more than 16 fresh ids (strict kernel) against eight; occupancy knobs on the new kernel
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
CFG=sift1m tools/variant_ab.sh r06_c7_sift deep10k:10000:X=1 nodeep10k:10000:LIB=lib_nodeep.so deep10kb:10000:X=1 nodeep10kb:10000:LIB=lib_nodeep.so deep12k:12500:X=1 nodeep12k:12500:LIB=lib_nodeep.so deep100k:100000:X=1 nodeep100k:100000:LIB=lib_nodeep.so wg20:10000:HNSWGPU_STRICT_WG_PER_CU=20,HNSWGPU_CAND_LDS=256 wg12:10000:HNSWGPU_STRICT_WG_PER_CU=12 2>&1 | grep -v "^queries in flight\|^last finishers" | cut -c1-260
CFG=mnist784 tools/variant_ab.sh r06_c7_mnist deep:10000:X=1 nodeep:10000:LIB=lib_nodeep.so 2>&1 | grep -v "^queries in flight\|^last finishers" | cut -c1-260
