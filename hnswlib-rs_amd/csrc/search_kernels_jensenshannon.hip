// search_kernels_jensenshannon.hip -- instantiates the search / descent / literal-heap / construction / test kernels for DistJensenShannon.
#define HNSW_THIS_METRIC DIST_JENSENSHANNON
#define HNSW_KERNELSET_FN kernels_jensenshannon
#include "search_kernels.hpp"
#include "search_kernels.inc"
#include "search_launchers.inc"
