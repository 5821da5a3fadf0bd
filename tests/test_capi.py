"""The C-ABI library: loads, exports every symbol include/hnsw_mi355x.h declares, keeps the reference's
struct layouts, and its reference-compatible entry points (src/libext.rs names) work on the host side.
No compute call here needs a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import uniform

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}()]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_every_declared_symbol_is_exported(native):
    names = declared_functions()
    assert len(names) >= 40
    lib = C.CDLL(native.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the python binding knows each of them
    assert set(names) == set(native._native.SYMBOLS)


def test_reference_struct_layouts(native):
    N = native._native
    assert C.sizeof(N.Neighbour_api) == 16 and N.Neighbour_api.d.offset == 8          # src/libext.rs:64-71
    assert C.sizeof(N.Neighbourhood_api) == 16                                          # src/libext.rs:82-87
    assert C.sizeof(N.Vec_api_Neighbourhood) == 16                                      # src/libext.rs:58-62
    assert N.DescriptionFFI.ef.offset == 8 and C.sizeof(N.DescriptionFFI) == 64         # src/libext.rs:1121-1141


def test_reference_style_load_dump_description(native, oracle, tmp_path, monkeypatch):
    """get_hnswio / load_hnswdump_f32_DistL2 / file_dump_f32 / load_hnsw_description / drop_hnsw_f32."""
    lib = native.lib()
    o = oracle.OracleHnsw(9, 300, 16, 30, "DistL2")
    o.insert_batch(uniform(300, 6, 8))
    o.file_dump(tmp_path, "cmp")
    monkeypatch.chdir(tmp_path)  # the reference's FFI always uses directory "." (src/libext.rs:31)
    io = lib.get_hnswio(3, b"cmp")
    assert io
    assert not lib.load_hnswdump_f32_DistCosine(io)  # null on a distance mismatch (src/libext.rs:298-301)
    api = lib.load_hnswdump_f32_DistL2(io)
    assert api
    idx = lib.hnswgpu_from_api(api)
    assert lib.hnswgpu_nb_point(idx) == 300 and lib.hnswgpu_dimension(idx) == 6
    assert lib.file_dump_f32(api, 4, b"cmp2") == 1
    from conftest import same_dump_after_reload
    assert same_dump_after_reload("cmp.hnsw.graph", "cmp2.hnsw.graph")
    assert open("cmp2.hnsw.data", "rb").read() == open("cmp.hnsw.data", "rb").read()
    d = lib.load_hnsw_description(len(b"cmp.hnsw.graph"), b"cmp.hnsw.graph")
    assert d and d.contents.max_nb_connection == 9 and d.contents.ef == 30 and d.contents.data_dimension == 6
    assert d.contents.dumpmode == 1 and d.contents.nb_point == 0  # reference quirks (src/libext.rs:1198-1206)
    assert C.string_at(d.contents.distname, d.contents.distname_len) == b"anndists::dist::distances::DistL2"
    lib.hnswgpu_free_description(d)
    assert not lib.load_hnsw_description(7, b"nothere")
    lib.drop_hnsw_f32(api)
    lib.hnswgpu_free_hnswio(io)


def test_reference_style_construction(native, oracle, tmp_path, monkeypatch):
    """init_hnsw_f32 + insert_f32 (serial) reproduces the oracle graph; parallel_insert_f32 accepts pointers."""
    lib = native.lib()
    monkeypatch.chdir(tmp_path)
    X = uniform(400, 5, 12)
    assert not lib.init_hnsw_f32(8, 40, 10, b"DistCosine")  # no DistCosine arm in the reference (src/libext.rs:468-523)
    assert not lib.init_hnsw_f32(8, 40, 7, b"DistFoo")
    api = lib.init_hnsw_f32(8, 40, 6, b"DistL2")
    assert api
    for i in range(400):
        lib.insert_f32(api, 5, X[i].ctypes.data, i)
    assert lib.file_dump_f32(api, 3, b"ins") == 1
    o = oracle.OracleHnsw(8, 400, 16, 40, "DistL2")
    o.insert_batch(X)
    o.file_dump(tmp_path, "orc")
    assert open("ins.hnsw.graph", "rb").read() == open("orc.hnsw.graph", "rb").read()
    lib.drop_hnsw_f32(api)
    api = lib.new_hnsw_f32(8, 40, 6, b"DistL2", 400, 16)
    ptrs = (C.c_void_p * 400)(*[X[i].ctypes.data for i in range(400)])
    ids = np.arange(400, dtype=np.uintp)
    lib.parallel_insert_f32(api, 400, 5, ptrs, ids.ctypes.data)
    assert lib.hnswgpu_nb_point(lib.hnswgpu_from_api(api)) == 400
    lib.drop_hnsw_f32(api)


def test_search_without_a_gpu_fails_loudly(native, tmp_path):
    if native.lib().hnswgpu_device_count() > 0:
        pytest.skip("a GPU is present")
    h = native.Hnsw(8, 100, 16, 20, "DistL2")
    h.insert_serial(uniform(100, 4, 1))
    with pytest.raises(native.HnswError) as e:
        h.parallel_search(uniform(2, 4, 2), 3, 10)
    assert e.value.code == native._native.ERR_DEVICE
    with pytest.raises(native.HnswError):
        h.upload(0)


def test_arithmetic_switch_is_validated_without_a_device(native):
    """hnswgpu_set_arithmetic: the two documented modes are accepted on any handle (the setting travels to the replicas made
    later), anything else is refused; the SIMD-order evaluation itself needs the device and says so."""
    h = native.Hnsw(8, 100, 16, 20, "DistL2")
    h.insert_serial(uniform(100, 4, 1))
    h.set_arithmetic("simd8")
    h.set_arithmetic("scalar")
    lib, N = native.lib(), native._native
    assert lib.hnswgpu_set_arithmetic(h.handle, 7) == N.ERR_ARG
    assert lib.hnswgpu_set_arithmetic(None, 0) == N.ERR_ARG
    if lib.hnswgpu_device_count() == 0:
        with pytest.raises(native.HnswError) as e:
            native.eval_distance_matrix("DistL2", uniform(2, 8, 1), uniform(3, 8, 2), batch=2, arithmetic="simd8")
        assert e.value.code == N.ERR_DEVICE
    with pytest.raises(native.HnswError) as e:  # the probability distances have no SIMD-order variant
        native.eval_distance_matrix("DistHellinger", uniform(2, 8, 1), uniform(3, 8, 2), batch=2, arithmetic="simd8")
    assert e.value.code in (N.ERR_ARG, N.ERR_DEVICE)


def test_empty_index_search_returns_empty(native):
    h = native.Hnsw(8, 10, 16, 20, "DistL2")
    assert h.parallel_search(uniform(3, 4, 1), 2, 5) == [[], [], []]  # src/hnsw.rs:1498-1503


def test_plain_c_caller_of_the_reference_style_symbols(native, tmp_path):
    """include/hnsw_mi355x.h is C99; a C program builds, dumps, reloads and describes an index through the crate's
    own FFI symbol names (src/libext.rs) without touching the GPU."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "c", "caller.c")
    exe = str(tmp_path / "caller")
    libdir = os.path.join(root, "hnswlib-rs_amd")
    subprocess.run(["gcc", "-std=c11", "-D_DEFAULT_SOURCE", "-Wall", "-Wextra", "-I", os.path.join(root, "include"), src, "-o", exe,
                    "-L", libdir, "-lhnsw_mi355x", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    work = tmp_path / "work"
    work.mkdir()
    r = subprocess.run([exe, str(work)], capture_output=True, text=True)
    assert r.returncode == 0, f"exit {r.returncode}: {r.stdout} {r.stderr}"
    assert (work / "c_caller.hnsw.graph").exists() and (work / "c_caller.hnsw.data").exists()


def test_host_distance_callback_is_refused_with_a_reason(native):
    """init_hnsw_ptrdist_f32 (src/libext.rs:643-655) cannot be served by a device path: NULL + a message, never a crash."""
    import ctypes as C
    lib = native.lib()
    cb = C.CFUNCTYPE(C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_ulonglong)(lambda a, b, n: 0.0)
    assert not lib.init_hnsw_ptrdist_f32(16, 200, C.cast(cb, C.c_void_p))
    assert "callback" in native._native.last_error()


def test_round5_entry_points_check_their_arguments_without_a_device(native):
    """hnswgpu_gather_sharded_answers / hnswgpu_lane_lab: argument errors are reported as status codes (no device needed to find
    them, nothing aborts), and without a HIP device the calls fail with the device error -- there is no CPU fallback."""
    lib = native.lib()
    one = np.zeros(4, np.uint32)
    # null buffers / bad shapes
    assert lib.hnswgpu_gather_sharded_answers(None, 1, None, 10, None, None, None, None, None, 0, None, None, None, None, None, None) != 0
    devs = (C.c_int * 1)(0)
    cnt = (C.c_uint64 * 1)(4)
    nullp = (C.c_void_p * 1)(None)
    assert lib.hnswgpu_gather_sharded_answers(devs, 1, cnt, 10, nullp, nullp, None, None, nullp, 0, None, None, None, None, None, None) != 0
    assert "null" in native._native.last_error() or "bad" in native._native.last_error()
    assert lib.hnswgpu_lane_lab(0, 9, 0, 0, 0, one.ctypes.data, 1, None, 0, one.ctypes.data, 4) != 0          # no such mode
    assert lib.hnswgpu_lane_lab(0, 1, 3, 0, 0, one.ctypes.data, 1, None, 0, one.ctypes.data, 4) != 0          # 3 slots per lane
    assert lib.hnswgpu_lane_lab(0, 2, 1, 65, 0, one.ctypes.data, 1, None, 0, one.ctypes.data, 4) != 0         # ef beyond 64 entries
    assert lib.hnswgpu_lane_lab(0, 3, 11, 20, 11, one.ctypes.data, 1, None, 0, one.ctypes.data, 4) != 0       # restbits != idbits - (tbits - 3)
    assert lib.hnswgpu_lane_lab(0, 0, 8, 0, 0, None, 1, None, 0, one.ctypes.data, 4) != 0                     # null script
    if lib.hnswgpu_device_count() == 0:
        assert lib.hnswgpu_lane_lab(0, 0, 8, 0, 0, one.ctypes.data, 1, None, 0, one.ctypes.data, 4) != 0      # no device: an error, not a CPU run
