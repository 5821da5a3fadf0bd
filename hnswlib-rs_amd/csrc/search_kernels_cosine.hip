// search_kernels_cosine.hip -- instantiates the search / exact-replay / test kernels for DistCosine.
#define HNSW_THIS_METRIC DIST_COSINE
#include "search_kernels.hpp"
#include "search_kernels.inc"

namespace hnswgpu {
namespace {
constexpr int M = HNSW_THIS_METRIC;
using KernelFn = void (*)(DeviceIndexView, SearchArgs);

template <int TABLE, bool STRICT>
KernelFn pick_slots(int slots) {
    switch (slots) {
        case 1: return hnsw_search_kernel<M, 1, TABLE, STRICT>;
        case 2: return hnsw_search_kernel<M, 2, TABLE, STRICT>;
        case 4: return hnsw_search_kernel<M, 4, TABLE, STRICT>;
        default: return hnsw_search_kernel<M, 16, TABLE, STRICT>;
    }
}
KernelFn pick(int slots, int table, bool strict) {
    if (table == TABLE_GLOBAL_BITMAP) return pick_slots<TABLE_GLOBAL_BITMAP, false>(slots);
    if (table == TABLE_LDS_CELL16) return strict ? pick_slots<TABLE_LDS_CELL16, true>(slots) : pick_slots<TABLE_LDS_CELL16, false>(slots);
    return strict ? pick_slots<TABLE_LDS_CELL32, true>(slots) : pick_slots<TABLE_LDS_CELL32, false>(slots);
}
hipError_t launch_search(int slots, int table, bool strict, uint32_t grid, size_t lds, hipStream_t stream,
                         const DeviceIndexView& ix, const SearchArgs& a) {
    hipLaunchKernelGGL(pick(slots, table, strict), dim3(grid), dim3(64), lds, stream, ix, a);
    return hipGetLastError();
}
hipError_t occupancy(int slots, int table, bool strict, size_t lds, int* per_cu) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, pick(slots, table, strict), 64, lds);
}
hipError_t launch_exact(int ns, uint32_t grid, size_t lds, hipStream_t stream, const DeviceIndexView& ix,
                        const SearchArgs& a, const ExactArgs& x) {
    if (ns == 1) hipLaunchKernelGGL((hnsw_search_exact_kernel<M, 1>), dim3(grid), dim3(64), lds, stream, ix, a, x);
    else if (ns == 2) hipLaunchKernelGGL((hnsw_search_exact_kernel<M, 2>), dim3(grid), dim3(64), lds, stream, ix, a, x);
    else hipLaunchKernelGGL((hnsw_search_exact_kernel<M, 0>), dim3(grid), dim3(64), lds, stream, ix, a, x);
    return hipGetLastError();
}
hipError_t launch_eval_pairs(uint32_t blocks, const float* a, const float* b, float* out, uint32_t n, uint32_t row_stride) {
    hipLaunchKernelGGL(eval_pairs_kernel<M>, dim3(blocks), dim3(64), 0, 0, a, b, out, n, row_stride);
    return hipGetLastError();
}
}  // namespace

const KernelSet& kernels_cosine() {
    static const KernelSet k{launch_search, occupancy, launch_exact, launch_eval_pairs};
    return k;
}
}  // namespace hnswgpu
