"""Builds and runs tests/cpp/test_hnsw_rs.cpp: the reference's own test scenarios written against the C++
mirror of the crate's interface (include/hnsw_rs.hpp) on top of the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def cpp_binary(native, tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "test_hnsw_rs"
    lib_dir = os.path.dirname(native.LIB_PATH)
    cmd = ["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_hnsw_rs.cpp"),
           "-o", str(out), native.LIB_PATH, f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True)
    return str(out)


def test_cpp_mirror_host_side(cpp_binary, tmp_path):
    r = subprocess.run([cpp_binary, "cpu", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpu mode OK" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_search(cpp_binary, tmp_path):
    r = subprocess.run([cpp_binary, "gpu", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu mode OK" in r.stdout


def test_worker_pool_stress(tmp_path):
    """csrc/worker_pool.hpp by itself (host-only): exactly-once task execution over thousands of sections, nested and
    concurrent sections without deadlock."""
    out = tmp_path / "test_worker_pool"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "hnswlib-rs_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "test_worker_pool.cpp"), "-o", str(out)], check=True, capture_output=True)
    r = subprocess.run([str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "worker pool OK" in r.stdout, r.stdout + r.stderr
