"""hnswlib-rs_amd -- MI355X-native batched HNSW search, drop-in for hnsw_rs's parallel_search path.

The directory name follows the reference repo ("hnswlib-rs") and is not a valid Python identifier;
import it through the root-level shim:  `import hnsw_rs_amd`.
"""
from ._native import build_native, lib, LIB_PATH, DIST, DIST_NAME  # noqa: F401
from .api import (reload_env, BatchResult, DataMap, Hnsw, HnswError, HnswIo, Neighbour, eval_distance_matrix, eval_distances,  # noqa: F401
                  load_description)
