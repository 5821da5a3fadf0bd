"""Import shim: exposes the package directory `hnswlib-rs_amd/` as the module `hnsw_rs_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hnswlib-rs_amd")
_spec = importlib.util.spec_from_file_location("hnsw_rs_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hnsw_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
