// search_kernels_jeffreys.hip -- instantiates the search / descent / literal-heap / construction / test kernels for DistJeffreys.
#define HNSW_THIS_METRIC DIST_JEFFREYS
#define HNSW_KERNELSET_FN kernels_jeffreys
#include "search_kernels.hpp"
#include "search_kernels.inc"
#include "search_launchers.inc"
