// search_kernels_hellinger.hip -- instantiates the search / descent / literal-heap / construction / test kernels for DistHellinger.
#define HNSW_THIS_METRIC DIST_HELLINGER
#define HNSW_KERNELSET_FN kernels_hellinger
#include "search_kernels.hpp"
#include "search_kernels.inc"
#include "search_launchers.inc"
