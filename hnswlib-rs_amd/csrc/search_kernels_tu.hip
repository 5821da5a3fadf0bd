// search_kernels_tu.hip -- the kernels of ONE metric, compiled three times per metric so that a clean build spreads
// over the cores:  -DHNSW_THIS_METRIC=<Dist id 0..6, or 7..10: the SIMD-order variants>  -DHNSW_PART=<0|1|2>
//   part 0: the strict search kernels (sorted-array loop + literal-heap continuation)
//   part 1: the lean search kernels (LDS tables and the HBM bitmap)
//   part 2: literal-heap kernel, descent / order kernels, construction searches, arithmetic test kernel, the KernelSet; the L2
//           unit of this part also carries the metric-independent helper kernels
#if !defined(HNSW_THIS_METRIC) || !defined(HNSW_PART)
#error "compile with -DHNSW_THIS_METRIC=<0..10> -DHNSW_PART=<0..2> (see the Makefile)"
#endif
#if HNSW_THIS_METRIC == 0 && HNSW_PART == 2
#define HNSW_SHARED_HELPERS 1
#endif
#include "search_kernels.hpp"
#include "search_kernels.inc"
#include "search_launchers.inc"
