#!/bin/bash
# round 6, call 4: the distance chain's DPP adds without the compiler's s_nop pairs -- arithmetic + parity tests, then against round 5's library
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -3
CFG=sift1m tools/variant_ab.sh r06_c4_sift new10k:10000:X=1 old10k:10000:LIB=lib_r05.so new10kb:10000:X=1 new12k:12500:X=1 old12k:12500:LIB=lib_r05.so new100k:100000:X=1 old100k:100000:LIB=lib_r05.so 2>&1 | grep -v "^queries in flight\|^last finishers" | cut -c1-260
CFG=mnist784 tools/variant_ab.sh r06_c4_mnist new:10000:X=1 old:10000:LIB=lib_r05.so 2>&1 | grep -v "^queries in flight\|^last finishers" | cut -c1-260
CFG=glove25_dot tools/variant_ab.sh r06_c4_dot new:10000:X=1 old:10000:LIB=lib_r05.so 2>&1 | grep -v "^queries in flight\|^last finishers" | cut -c1-260
