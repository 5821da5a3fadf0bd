// search_kernels_dot.hip -- instantiates the search / descent / exact-replay / test kernels for DistDot.
#define HNSW_THIS_METRIC DIST_DOT
#define HNSW_KERNELSET_FN kernels_dot
#include "search_kernels.hpp"
#include "search_kernels.inc"
#include "search_launchers.inc"
