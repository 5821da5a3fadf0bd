"""tools/asan_host_suite.sh builds the host side of the C ABI against tests/cpp/stub_device.cpp (a stand-in for search_device.hip that
reports "no device").  The stand-in has to define every device entry the host sources reference: this test links the same sources
with -Wl,-z,defs (no sanitizer, -O0) so that a new device entry without its stand-in fails HERE and not when the sanitizer run is
next attempted.  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stub_device_defines_every_device_entry_of_the_host_sources(tmp_path):
    c = os.path.join(ROOT, "hnswlib-rs_amd", "csrc")
    out = tmp_path / "libhnsw_stub.so"
    cmd = ["g++", "-O0", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-I" + c, "-I" + os.path.join(ROOT, "include"), "-shared",
           "-Wl,-z,defs", "-o", str(out)] + [os.path.join(c, f) for f in ("capi.cpp", "builder.cpp", "hnswio.cpp", "datamap.cpp")] + \
          [os.path.join(ROOT, "tests", "cpp", "stub_device.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
