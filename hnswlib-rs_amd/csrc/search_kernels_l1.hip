// search_kernels_l1.hip -- instantiates the search / descent / exact-replay / test kernels for DistL1.
#define HNSW_THIS_METRIC DIST_L1
#define HNSW_KERNELSET_FN kernels_l1
#include "search_kernels.hpp"
#include "search_kernels.inc"
#include "search_launchers.inc"
