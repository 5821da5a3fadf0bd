"""include/hnsw_mi355x.h is the single source of truth of the C ABI: the ctypes binding (hnswlib-rs_amd/_native.py) is
GENERATED from it at import time (hnswlib-rs_amd/_cheader.py).  These tests pin the generator: struct layouts against
what the C compiler computes from the same header, prototypes against the header text, and refusal of anything it does
not understand (no silent guesses)."""
import ctypes as C
import os
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "hnsw_mi355x.h")


def test_struct_layouts_equal_the_c_compilers(native, tmp_path):
    hdr = native._native.HEADER
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for sname, cls in hdr.structs.items():
        lines.append(f'  printf("{sname} %zu\\n", sizeof({sname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{sname}.{fname} %zu\\n", offsetof({sname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert len(hdr.structs) >= 6
    for sname, cls in hdr.structs.items():
        assert int(got[sname]) == C.sizeof(cls), sname
        for fname, _ in cls._fields_:
            assert int(got[f"{sname}.{fname}"]) == getattr(cls, fname).offset, f"{sname}.{fname}"


def test_bindings_follow_the_header_text(native):
    """Editing a prototype in the header alone changes the binding: there is no second, hand-kept table."""
    ch = native._native._cheader
    text = open(HEADER).read()
    base = ch.Header(text)
    assert base.prototypes.keys() == native._native.SYMBOLS.keys() and len(base.prototypes) >= 62
    assert "int hnswgpu_upload(hnswgpu_index* idx, int device);" in text
    edited = ch.Header(text.replace("int hnswgpu_upload(hnswgpu_index* idx, int device);",
                                    "int64_t hnswgpu_upload(hnswgpu_index* idx, int device, uint32_t flags);"))
    res, args, names, _ = edited.prototypes["hnswgpu_upload"]
    assert res is C.c_int64 and args == [C.c_void_p, C.c_int, C.c_uint32] and names == ["idx", "device", "flags"]
    assert base.prototypes["hnswgpu_upload"][:2] == (C.c_int, [C.c_void_p, C.c_int])
    # structs too: a field added to the header's typedef moves the ctypes layout
    edited = ch.Header(text.replace("    uint64_t gpu_window;", "    uint64_t gpu_window;\n    uint32_t extra;", 1))
    assert C.sizeof(edited.structs["hnswgpu_build_params"]) == C.sizeof(base.structs["hnswgpu_build_params"]) + 8
    # what the generator does not understand is an error at import, never a guess
    with pytest.raises(ValueError):
        ch.Header(text.replace("int hnswgpu_upload(hnswgpu_index* idx, int device);", "int hnswgpu_upload(hnswgpu_index* idx, long double device);"))


def test_every_prototype_is_exported_with_the_generated_signature(native):
    lib = native.lib()
    hdr = native._native.HEADER
    for name, (res, args, _names, text) in hdr.prototypes.items():
        fn = getattr(lib, name)
        assert fn.restype is res and list(fn.argtypes) == list(args), text
    # the reference-compatible symbols keep the reference's argument counts (src/libext.rs:205-254, :458-523)
    assert len(hdr.prototypes["parallel_search_neighbours_f32"][1]) == 6
    assert len(hdr.prototypes["search_neighbours_f32"][1]) == 5
    assert hdr.prototypes["parallel_search_neighbours_f32"][0] == C.POINTER(hdr.structs["Vec_api_Neighbourhood"])


def test_every_exported_symbol_has_a_parsed_prototype(native):
    """The other direction: whatever the library EXPORTS under the ABI's names (hnswgpu_* and the reference's own f32
    symbols) has a prototype the header parser understood -- a declaration the regex parser skipped silently would show up
    here as an exported symbol without a binding."""
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    ref_names = {"get_hnswio", "init_hnsw_f32", "new_hnsw_f32", "insert_f32", "parallel_insert_f32", "search_neighbours_f32",
                 "parallel_search_neighbours_f32", "file_dump_f32", "drop_hnsw_f32", "load_hnsw_description", "init_rust_log"}
    abi = {s for s in exported if s.startswith("hnswgpu_") or s.startswith("load_hnswdump_f32_") or s in ref_names}
    assert len(abi) >= 60
    missing = sorted(abi - set(native._native.SYMBOLS))
    assert not missing, f"exported without a parsed prototype in include/hnsw_mi355x.h: {missing}"
    # and the package's own copy of the header (what a relocated package binds from) is the tree's
    pkg = os.path.join(os.path.dirname(native.LIB_PATH), "hnsw_mi355x.h")
    assert os.path.exists(pkg) and open(pkg).read() == open(HEADER).read()
