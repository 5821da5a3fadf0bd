#!/bin/bash
# The host side of the C ABI (capi.cpp, builder.cpp, hnswio.cpp, datamap.cpp) built with AddressSanitizer + UBSan against a
# stand-in for the device (tests/cpp/stub_device.cpp: every device entry reports "no device", like the product library on a
# box without a GPU), then the CPU test-suite run against that library.  No GPU needed; ~3 minutes.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/hnswlib-rs_amd/csrc
OUT=${1:-/tmp/hnsw_asan}
mkdir -p $OUT
g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -I$C -I$ROOT/include -shared -o $OUT/libhnsw_asan.so $C/capi.cpp $C/builder.cpp $C/hnswio.cpp $C/datamap.cpp $ROOT/tests/cpp/stub_device.cpp
cd $ROOT
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
HNSW_MI355X_LIB=$OUT/libhnsw_asan.so python -m pytest tests -q -m "not gpu" -p no:cacheprovider --ignore=tests/test_cpp_mirror.py
