#!/bin/bash
# round 6, call 5: asm chain + micro-diet (scalar length masks, fill count carried into the insertion, straight-line first CAS), hooks read once: GPU suite, then against round 5's library
cd "$(dirname "$0")/.."
tools/gpu_call.sh suite 2>&1 | tail -8
CFG=sift1m tools/variant_ab.sh r06_c5_sift new10k:10000:X=1 old10k:10000:LIB=lib_r05.so new10kb:10000:X=1 new12k:12500:X=1 old12k:12500:LIB=lib_r05.so new100k:100000:X=1 old100k:100000:LIB=lib_r05.so 2>&1 | grep -v "^queries in flight\|^last finishers" | cut -c1-260
CFG=glove25 tools/variant_ab.sh r06_c5_glove new:10000:X=1 old:10000:LIB=lib_r05.so 2>&1 | grep -v "^queries in flight\|^last finishers" | cut -c1-260
