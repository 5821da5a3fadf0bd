// search_kernels_cosine.hip -- instantiates the search / descent / exact-replay / test kernels for DistCosine.
#define HNSW_THIS_METRIC DIST_COSINE
#define HNSW_KERNELSET_FN kernels_cosine
#include "search_kernels.hpp"
#include "search_kernels.inc"
#include "search_launchers.inc"
