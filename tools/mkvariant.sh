#!/bin/bash
# Builds a kernel A/B variant next to the product library: every translation unit recompiled with extra -D flags
# (some experiments change structs shared by host and device code), objects kept out of the source tree.
#   tools/mkvariant.sh NAME "-DHNSW_LB_WAVES=4 ..."   ->  hnswlib-rs_amd/lib_NAME.so
# Run it on the GPU box side by side with the default build in ONE gpurun call (the graph differs per box):
#   HNSW_MI355X_LIB=$PWD/hnswlib-rs_amd/lib_NAME.so python bench.py --no-cpu-baseline --no-recall | python tools/bench_line.py
# Variant libraries are git-ignored (*.so) but travel with gpurun: delete them when done.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/hnswlib-rs_amd/csrc"
OBJ=/tmp/hnsw_variant_$1
mkdir -p $OBJ
CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread"
for m in hnswio builder datamap capi; do
  g++ $CXXFLAGS $2 -c $m.cpp -o $OBJ/$m.o &
done
for m in search_kernels_l2 search_kernels_cosine search_kernels_dot search_kernels_l1 search_kernels_hellinger search_kernels_jeffreys search_kernels_jensenshannon search_device; do
  /opt/rocm/bin/hipcc $CXXFLAGS --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt $2 -c $m.hip -o $OBJ/$m.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o ../lib_$1.so $OBJ/hnswio.o $OBJ/builder.o $OBJ/datamap.o $OBJ/capi.o \
    $OBJ/search_device.o $OBJ/search_kernels_l2.o $OBJ/search_kernels_cosine.o $OBJ/search_kernels_dot.o $OBJ/search_kernels_l1.o $OBJ/search_kernels_hellinger.o $OBJ/search_kernels_jeffreys.o $OBJ/search_kernels_jensenshannon.o \
    -Wl,-rpath,/opt/rocm/lib
ls -la ../lib_$1.so
