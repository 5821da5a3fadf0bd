#!/bin/bash
# Builds a kernel A/B variant next to the product library: every translation unit recompiled with extra -D flags
# (some experiments change structs shared by host and device code), objects kept out of the source tree.
#   tools/mkvariant.sh NAME "-DHNSW_LB_WAVES=4 ..."   ->  hnswlib-rs_amd/lib_NAME.so
# Run it on the GPU box side by side with the default build in ONE gpurun call (the graph differs per box):
#   HNSW_MI355X_LIB=$PWD/hnswlib-rs_amd/lib_NAME.so python bench.py --no-cpu-baseline --no-recall | python tools/bench_line.py
# Variant libraries are git-ignored (*.so) but travel with gpurun: delete them when done.
# ONLY_METRICS="0 1" recompiles just those metrics' kernel units (Dist ids) and takes every other object from the
# product build in csrc/ -- for flags that only touch device code of search_kernels.inc (minutes saved).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/hnswlib-rs_amd/csrc"
OBJ=/tmp/hnsw_variant_$1
mkdir -p $OBJ
CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread"
HIPFLAGS="--offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt"
if [ -n "${ONLY_METRICS:-}" ]; then
  flock .build.lock make -j8 >/dev/null   # the product's objects, up to date (one make at a time)
  for f in hnswio builder datamap capi search_device; do cp $f.o $OBJ/$f.o; done
  for m in 0 1 2 3 4 5 6 7 8 9 10; do for p in 0 1 2; do cp sk_${m}_$p.o $OBJ/sk_${m}_$p.o; done; done
  METRICS="$ONLY_METRICS"
else
  for m in hnswio builder datamap capi; do
    g++ $CXXFLAGS $2 -c $m.cpp -o $OBJ/$m.o &
  done
  /opt/rocm/bin/hipcc $CXXFLAGS $HIPFLAGS $2 -c search_device.hip -o $OBJ/search_device.o &
  METRICS="0 1 2 3 4 5 6 7 8 9 10"
fi
for m in $METRICS; do
  for p in 0 1 2; do
    /opt/rocm/bin/hipcc $CXXFLAGS $HIPFLAGS $2 -DHNSW_THIS_METRIC=$m -DHNSW_PART=$p -c search_kernels_tu.hip -o $OBJ/sk_${m}_$p.o &
  done
  wait   # three units of one metric at a time next to the host objects: bounded memory
done
wait
KOBJS=""
for m in 0 1 2 3 4 5 6 7 8 9 10; do for p in 0 1 2; do KOBJS="$KOBJS $OBJ/sk_${m}_$p.o"; done; done
/opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o ../lib_$1.so $OBJ/hnswio.o $OBJ/builder.o $OBJ/datamap.o $OBJ/capi.o \
    $OBJ/search_device.o $KOBJS -Wl,-rpath,/opt/rocm/lib
ls -la ../lib_$1.so
