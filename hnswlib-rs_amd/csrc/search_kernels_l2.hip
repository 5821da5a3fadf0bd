// search_kernels_l2.hip -- instantiates the search / descent / exact-replay / test kernels for DistL2.
#define HNSW_THIS_METRIC DIST_L2
#define HNSW_SHARED_HELPERS 1  // this unit also carries the launchers of the metric-independent helper kernels
#define HNSW_KERNELSET_FN kernels_l2
#include "search_kernels.hpp"
#include "search_kernels.inc"
#include "search_launchers.inc"
