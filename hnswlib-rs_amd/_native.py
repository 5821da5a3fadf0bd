"""Loader of libhnsw_mi355x.so (the C ABI of include/hnsw_mi355x.h) through ctypes.

The library is built in-tree by `build_native()` (hipcc --offload-arch=gfx950).  If it is missing the
import of the binding fails loudly: there is no Python / CPU fallback for the search path.
"""
import ctypes as C
import os
import subprocess

from . import _cheader

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libhnsw_mi355x.so")
# tuning hook: load a differently-built variant of the same library (kernel A/B runs in one process tree)
LIB_OVERRIDE = os.environ.get("HNSW_MI355X_LIB")

# the header is looked for next to the package (an installed or relocated copy ships it as package data: build_native()
# refreshes that copy) and, in the source tree, under ../include -- the tree's copy wins when both exist
_TREE_HEADER = os.path.join(os.path.dirname(PKG_DIR), "include", "hnsw_mi355x.h")
_PKG_HEADER = os.path.join(PKG_DIR, "hnsw_mi355x.h")
HEADER_PATH = _TREE_HEADER if os.path.exists(_TREE_HEADER) else _PKG_HEADER
# The header is the one description of the C ABI; structures, prototypes and codes below are READ FROM IT (._cheader).
HEADER = _cheader.load(HEADER_PATH)
OK, ERR_ARG, ERR_IO, ERR_FORMAT, ERR_DISTANCE, ERR_TYPE, ERR_DEVICE, ERR_EMPTY, ERR_REF_PANIC = (
    HEADER.constants["HNSWGPU_" + n] for n in ("OK", "ERR_ARG", "ERR_IO", "ERR_FORMAT", "ERR_DISTANCE", "ERR_TYPE", "ERR_DEVICE",
                                               "ERR_EMPTY", "ERR_REF_PANIC"))
DIST = {"DistL2": HEADER.constants["HNSWGPU_DIST_L2"], "DistCosine": HEADER.constants["HNSWGPU_DIST_COSINE"],
        "DistDot": HEADER.constants["HNSWGPU_DIST_DOT"], "DistL1": HEADER.constants["HNSWGPU_DIST_L1"],
        "DistHellinger": HEADER.constants["HNSWGPU_DIST_HELLINGER"], "DistJeffreys": HEADER.constants["HNSWGPU_DIST_JEFFREYS"],
        "DistJensenShannon": HEADER.constants["HNSWGPU_DIST_JENSENSHANNON"]}
DIST_NAME = {v: k for k, v in DIST.items()}


def build_native(force=False, verbose=False):
    """Compile every HIP/C++ source for gfx950 into libhnsw_mi355x.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR)
            if f.endswith((".cpp", ".hip", ".hpp", ".inc")) or f == "Makefile"]
    srcs.append(HEADER_PATH)

    def fresh():
        return (os.path.exists(LIB_PATH)
                and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs))

    def ship_header():  # the package's own copy of the header (what a relocated package binds from)
        if os.path.exists(_TREE_HEADER):
            try:
                if not os.path.exists(_PKG_HEADER) or open(_PKG_HEADER, "rb").read() != open(_TREE_HEADER, "rb").read():
                    with open(_PKG_HEADER, "wb") as f:
                        f.write(open(_TREE_HEADER, "rb").read())
            except OSError:
                pass

    if not force and fresh():
        ship_header()
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
    ship_header()
    return LIB_PATH


# the header's structures (ctypes.Structure classes generated from its typedefs)
Description = HEADER.structs["hnswgpu_description"]
BuildParams = HEADER.structs["hnswgpu_build_params"]
Neighbour_api = HEADER.structs["Neighbour_api"]                    # src/libext.rs:64-71
Neighbourhood_api = HEADER.structs["Neighbourhood_api"]            # src/libext.rs:82-87
Vec_api_Neighbourhood = HEADER.structs["Vec_api_Neighbourhood"]    # src/libext.rs:58-62
DescriptionFFI = HEADER.structs["DescriptionFFI"]                  # src/libext.rs:1121-1141

# every symbol include/hnsw_mi355x.h declares: name -> (restype, argtypes)
SYMBOLS = {name: (res, args) for name, (res, args, _names, _text) in HEADER.prototypes.items()}

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
