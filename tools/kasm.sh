#!/bin/bash
# Device assembly of one kernel translation unit: tools/kasm.sh METRIC PART OUT.s [extra flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/hnswlib-rs_amd/csrc"
M=$1; P=$2; O=$3; shift 3
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt "$@" \
  -DHNSW_THIS_METRIC=$M -DHNSW_PART=$P --cuda-device-only -S search_kernels_tu.hip -o $O 2>/dev/null
