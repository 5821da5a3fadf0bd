"""Loader of libhnsw_mi355x.so (the C ABI of include/hnsw_mi355x.h) through ctypes.

The library is built in-tree by `build_native()` (hipcc --offload-arch=gfx950).  If it is missing the
import of the binding fails loudly: there is no Python / CPU fallback for the search path.
"""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libhnsw_mi355x.so")
# tuning hook: load a differently-built variant of the same library (kernel A/B runs in one process tree)
LIB_OVERRIDE = os.environ.get("HNSW_MI355X_LIB")

OK, ERR_ARG, ERR_IO, ERR_FORMAT, ERR_DISTANCE, ERR_TYPE, ERR_DEVICE, ERR_EMPTY, ERR_REF_PANIC = range(9)
DIST = {"DistL2": 0, "DistCosine": 1, "DistDot": 2, "DistL1": 3, "DistHellinger": 4, "DistJeffreys": 5, "DistJensenShannon": 6}
DIST_NAME = {v: k for k, v in DIST.items()}


def build_native(force=False, verbose=False):
    """Compile every HIP/C++ source for gfx950 into libhnsw_mi355x.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR)
            if f.endswith((".cpp", ".hip", ".hpp", ".inc")) or f == "Makefile"]
    srcs.append(os.path.join(os.path.dirname(PKG_DIR), "include", "hnsw_mi355x.h"))

    def fresh():
        return (os.path.exists(LIB_PATH)
                and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs))

    if not force and fresh():
        return LIB_PATH
    # one process per GPU may get here at the same time (bench.py --gpus N): build under a lock, the others then
    # find the library up to date
    import fcntl
    with open(os.path.join(CSRC_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not fresh():
                r = subprocess.run(["make", "-C", CSRC_DIR, "-j%d" % max(2, os.cpu_count() or 4)], capture_output=True, text=True)
                if verbose or r.returncode != 0:
                    print(r.stdout)
                    print(r.stderr)
                if r.returncode != 0:
                    raise RuntimeError("building libhnsw_mi355x.so failed")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


class Description(C.Structure):
    _fields_ = [("format_version", C.c_uint32), ("dumpmode", C.c_uint8), ("max_nb_connection", C.c_uint8),
                ("nb_layer", C.c_uint8), ("level_scale", C.c_double), ("ef_construction", C.c_uint64),
                ("nb_point", C.c_uint64), ("dimension", C.c_uint64), ("distname", C.c_char * 260),
                ("t_name", C.c_char * 260)]


class BuildParams(C.Structure):
    _fields_ = [("max_nb_connection", C.c_uint64), ("ef_construction", C.c_uint64), ("max_layer", C.c_uint64),
                ("dist", C.c_int), ("level_scale_factor", C.c_double), ("extend_candidates", C.c_int),
                ("keep_pruned", C.c_int), ("nthreads", C.c_int), ("fast_arithmetic", C.c_int), ("gpu_assist", C.c_int),
                ("gpu_device", C.c_int), ("gpu_window", C.c_uint64)]


class Neighbour_api(C.Structure):  # src/libext.rs:64-71
    _fields_ = [("id", C.c_size_t), ("d", C.c_float)]


class Neighbourhood_api(C.Structure):  # src/libext.rs:82-87
    _fields_ = [("nbgh", C.c_int64), ("neighbours", C.POINTER(Neighbour_api))]


class Vec_api_Neighbourhood(C.Structure):  # src/libext.rs:58-62
    _fields_ = [("len", C.c_int64), ("ptr", C.POINTER(Neighbourhood_api))]


class DescriptionFFI(C.Structure):  # src/libext.rs:1121-1141
    _fields_ = [("dumpmode", C.c_uint8), ("max_nb_connection", C.c_uint8), ("nb_layer", C.c_uint8),
                ("ef", C.c_size_t), ("nb_point", C.c_size_t), ("data_dimension", C.c_size_t),
                ("distname_len", C.c_size_t), ("distname", C.c_void_p), ("t_name_len", C.c_size_t),
                ("t_name", C.c_void_p)]


# every symbol include/hnsw_mi355x.h declares: (restype, argtypes)
_VP, _U64, _SZ, _I = C.c_void_p, C.c_uint64, C.c_size_t, C.c_int
SYMBOLS = {
    "hnswgpu_last_error": (C.c_char_p, []),
    "hnswgpu_load_dump": (_I, [C.c_char_p, C.c_char_p, _I, C.POINTER(_VP)]),
    "hnswgpu_file_dump": (_I, [_VP, C.c_char_p, C.c_char_p]),
    "hnswgpu_free_index": (None, [_VP]),
    "hnswgpu_load_description": (_I, [C.c_char_p, C.POINTER(Description)]),
    "hnswgpu_get_description": (_I, [_VP, C.POINTER(Description)]),
    "hnswgpu_datamap_open": (_I, [C.c_char_p, C.c_char_p, C.POINTER(_VP)]),
    "hnswgpu_datamap_close": (None, [_VP]),
    "hnswgpu_datamap_get_data": (_VP, [_VP, _U64]),
    "hnswgpu_datamap_nb_data": (_U64, [_VP]),
    "hnswgpu_datamap_dimension": (_U64, [_VP]),
    "hnswgpu_datamap_distname": (C.c_char_p, [_VP]),
    "hnswgpu_datamap_typename": (C.c_char_p, [_VP]),
    "hnswgpu_datamap_ids": (_U64, [_VP, _VP, _U64]),
    "hnswgpu_build": (_I, [_VP, _U64, _U64, _VP, C.POINTER(BuildParams), C.POINTER(_VP)]),
    "hnswgpu_insert": (_I, [_VP, _VP, _U64, _U64, _VP, _I]),
    "hnswgpu_insert_gpu": (_I, [_VP, _VP, _U64, _U64, _VP, _I, _I, _U64]),
    "hnswgpu_nb_point": (_U64, [_VP]),
    "hnswgpu_dimension": (_U64, [_VP]),
    "hnswgpu_dist": (_I, [_VP]),
    "hnswgpu_layer_nb_point": (_U64, [_VP, C.c_uint]),
    "hnswgpu_max_level_observed": (_I, [_VP]),
    "hnswgpu_entry_point": (_I, [_VP, _VP, _VP, _VP]),
    "hnswgpu_neighbours": (C.c_int64, [_VP, C.c_uint, C.c_int32, C.c_uint, _U64, _VP, _VP, _VP, _VP]),
    "hnswgpu_upload": (_I, [_VP, _I]),
    "hnswgpu_device_count": (_I, []),
    "hnswgpu_search_batch": (_I, [_VP, _VP, _U64, _U64, _U64, _U64, _VP, _VP, _VP, _VP, _VP]),
    "hnswgpu_search_batch_device": (_I, [_VP, _VP, _U64, _U64, _U64, _U64, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hnswgpu_search_batch_device_begin": (_I, [_VP, _VP, _U64, _U64, _U64, _U64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hnswgpu_search_batch_end": (_I, [_VP]),
    "hnswgpu_search_batch_filtered": (_I, [_VP, _VP, _U64, _U64, _U64, _U64, _VP, _U64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hnswgpu_search_batch_sharded": (_I, [_VP, _VP, _I, _VP, _U64, _U64, _U64, _U64, _VP, _VP, _VP, _VP, _VP]),
    "hnswgpu_search_batch_filtered_device": (_I, [_VP, _VP, _U64, _U64, _U64, _U64, _VP, _U64, _VP, _VP, _VP, _VP, _VP, _VP,
                                                  _VP, C.POINTER(C.c_uint32)]),
    "hnswgpu_last_kernel_ms": (_I, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "hnswgpu_last_search_kernel_ms": (_I, [_VP, C.POINTER(C.c_double)]),
    "hnswgpu_set_strict_ties": (_I, [_VP, _I]),
    "hnswgpu_last_tie_count": (_I, [_VP, C.POINTER(C.c_uint32)]),
    "hnswgpu_eval_distances": (_I, [_I, _VP, _VP, _U64, _U64, _VP]),
    "hnswgpu_eval_distance_matrix": (_I, [_I, _VP, _U64, _VP, _U64, _U64, C.c_uint32, _VP]),
    # reference-compatible symbols (src/libext.rs)
    "get_hnswio": (_VP, [_U64, C.c_char_p]),
    "load_hnswdump_f32_DistL1": (_VP, [_VP]),
    "load_hnswdump_f32_DistL2": (_VP, [_VP]),
    "load_hnswdump_f32_DistCosine": (_VP, [_VP]),
    "load_hnswdump_f32_DistDot": (_VP, [_VP]),
    "load_hnswdump_f32_DistJensenShannon": (_VP, [_VP]),
    "load_hnswdump_f32_DistJeffreys": (_VP, [_VP]),
    "init_hnsw_f32": (_VP, [_SZ, _SZ, _SZ, C.c_char_p]),
    "new_hnsw_f32": (_VP, [_SZ, _SZ, _SZ, C.c_char_p, _SZ, _SZ]),
    "init_hnsw_ptrdist_f32": (_VP, [_SZ, _SZ, _VP]),
    "insert_f32": (None, [_VP, _SZ, _VP, _SZ]),
    "parallel_insert_f32": (None, [_VP, _SZ, _SZ, _VP, _VP]),
    "search_neighbours_f32": (C.POINTER(Neighbourhood_api), [_VP, _SZ, _VP, _SZ, _SZ]),
    "parallel_search_neighbours_f32": (C.POINTER(Vec_api_Neighbourhood), [_VP, _SZ, C.c_int64, _VP, _SZ, _SZ]),
    "file_dump_f32": (C.c_int64, [_VP, _SZ, C.c_char_p]),
    "drop_hnsw_f32": (None, [_VP]),
    "load_hnsw_description": (C.POINTER(DescriptionFFI), [_SZ, C.c_char_p]),
    "init_rust_log": (None, []),
    "hnswgpu_free_neighbourhood": (None, [C.POINTER(Neighbourhood_api)]),
    "hnswgpu_free_neighbourhood_vec": (None, [C.POINTER(Vec_api_Neighbourhood)]),
    "hnswgpu_free_hnswio": (None, [_VP]),
    "hnswgpu_free_description": (None, [C.POINTER(DescriptionFFI)]),
    "hnswgpu_from_api": (_VP, [_VP]),
}

_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        path = LIB_OVERRIDE or LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with hnsw_rs_amd.build_native() "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().hnswgpu_last_error().decode(errors="replace")
