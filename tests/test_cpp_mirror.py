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


def _tsan_build(tmp_path, name, sources):
    """g++ -fsanitize=thread; skips when the toolchain has no ThreadSanitizer runtime or the box cannot run one (address-space
    layouts the runtime does not know make every instrumented program die at start-up)."""
    probe_src, probe = tmp_path / "tsan_probe.cpp", tmp_path / "tsan_probe"
    probe_src.write_text("#include <cstdio>\nint main() { std::puts(\"probe ok\"); return 0; }\n")
    r = subprocess.run(["g++", "-fsanitize=thread", "-pthread", str(probe_src), "-o", str(probe)], capture_output=True, text=True)
    if r.returncode != 0 or subprocess.run([str(probe)], capture_output=True, text=True).stdout.strip() != "probe ok":
        pytest.skip("ThreadSanitizer is not usable here")
    out = tmp_path / name
    csrc = os.path.join(ROOT, "hnswlib-rs_amd", "csrc")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-I", csrc, *sources, "-o", str(out)],
                       capture_output=True, text=True)
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr
    return str(out)


def test_host_builder_is_clean_under_thread_sanitizer(tmp_path):
    """parallel_insert on the host cores (csrc/builder.cpp): searches read neighbour lists without a lock while other threads
    rewrite them (EdgeList: a publish-once buffer behind an atomic pointer, reclaimed when the builder dies).  A data race
    there would be a torn list in a search; ThreadSanitizer must have nothing to report over two batches on 8 threads."""
    csrc = os.path.join(ROOT, "hnswlib-rs_amd", "csrc")
    exe = _tsan_build(tmp_path, "test_builder_tsan", [os.path.join(ROOT, "tests", "cpp", "test_builder_tsan.cpp"),
                                                      os.path.join(csrc, "builder.cpp"), os.path.join(csrc, "hnswio.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "builder under tsan OK" in r.stdout, r.stdout + r.stderr[-2000:]


def test_worker_pool_is_clean_under_thread_sanitizer(tmp_path):
    exe = _tsan_build(tmp_path, "test_worker_pool_tsan", [os.path.join(ROOT, "tests", "cpp", "test_worker_pool.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "worker pool OK" in r.stdout, r.stdout + r.stderr[-2000:]


def _window_logic_sources():
    csrc = os.path.join(ROOT, "hnswlib-rs_amd", "csrc")
    return [os.path.join(ROOT, "tests", "cpp", "test_builder_window_logic.cpp"), os.path.join(csrc, "builder.cpp"), os.path.join(csrc, "hnswio.cpp")]


def test_gpu_assisted_construction_host_side_against_a_mock_device(tmp_path):
    """GraphBuilder::insert_batch_gpu with the device replaced by a CPU mock that keeps the frozen snapshot and searches it with
    the builder's own semantics: window 1 == the serial insertion (dumps byte-identical), growing windows on several threads,
    a backend that fails at its third window (the host builder finishes, the call says so), a backend that refuses (index
    unchanged).  The device side of the same protocol is GPU-tested (tests/test_gpu_round2.py, tests/test_gpu_round3.py)."""
    exe = tmp_path / "test_builder_window_logic"
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-pthread", "-I", os.path.join(ROOT, "hnswlib-rs_amd", "csrc"),
                    *_window_logic_sources(), "-o", str(exe)], check=True, capture_output=True)
    out = tmp_path / "dumps"
    out.mkdir()
    r = subprocess.run([str(exe), str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "window logic OK" in r.stdout, r.stdout + r.stderr


def test_gpu_assisted_construction_host_side_is_clean_under_thread_sanitizer(tmp_path):
    exe = _tsan_build(tmp_path, "test_builder_window_logic_tsan", ["-ffp-contract=off", *_window_logic_sources()])
    out = tmp_path / "dumps"
    out.mkdir()
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=900)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "window logic OK" in r.stdout, r.stdout + r.stderr[-2000:]
