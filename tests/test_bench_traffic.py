"""bench.py's roofline.traffic leg on CPU: a stand-in for rocprofv3 writes the counter tables the real tool writes (one row per
dispatch, counter and -- as on hardware -- XCD instance), and live_traffic must average per dispatch over the STRICT launches of the
default arithmetic only, apply the guide's corrections (FETCH_SIZE in KB, x 2 on gfx950; WRITE_SIZE in KB), never mix the two passes,
and give up cleanly (None + a reason) when the tool is missing, fails, or this process is itself being profiled."""
import argparse
import os
import stat
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FAKE = textwrap.dedent('''\
    #!/usr/bin/env python3
    import os, sys
    a = sys.argv[1:]
    if os.environ.get("FAKE_ROCPROF_FAIL"):
        sys.stderr.write("boom"); sys.exit(3)
    ctrs = a[a.index("--pmc") + 1:a.index("-d")]
    out = a[a.index("-d") + 1]
    assert "--kernel-trace" not in a and "--stats" not in a          # counters are never mixed with trace domains
    assert "--no-traffic" in a and "--no-boundary" in a                # the child measures nothing else, and does not recurse
    d = os.path.join(out, "host", "1234")
    os.makedirs(d)
    K = "void hnswgpu::(anonymous namespace)::hnsw_search_kernel<%s>(hnswgpu::DeviceIndexView, hnswgpu::SearchArgs)"
    val = {"FETCH_SIZE": 1000.0, "TCC_EA0_RDREQ_sum": 16000.0, "WRITE_SIZE": 100.0}
    with open(os.path.join(d, "1234_counter_collection.csv"), "w") as f:
        f.write("Correlation_Id,Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\\n")
        disp = 0
        for targs, scale in (("0, 1, 0, true", 1.0), ("0, 1, 0, true", 3.0), ("0, 1, 0, false", 50.0), ("7, 1, 0, true", 70.0)):
            disp += 1
            for c in ctrs:
                for xcd in range(8):                                   # one row per XCD: summed per dispatch
                    f.write('%d,%d,"%s",%s,%f\\n' % (disp, disp, K % targs, c, val[c] * scale / 8))
        f.write('9,9,"order_desc_kernel(...)",%s,12345\\n' % ctrs[0])
''')


@pytest.fixture
def fake_tool(tmp_path, monkeypatch):
    p = tmp_path / "rocprofv3"
    p.write_text(FAKE)
    p.chmod(p.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ.get("PATH", ""))
    for k in list(os.environ):
        if k.startswith("ROCPROF"):
            monkeypatch.delenv(k)
    return p


def _args(tmp_path):
    return argparse.Namespace(config="random10k", data="uniform", cache_dir=str(tmp_path), batches=4, n=0, nq=0, ef=0)


def test_live_traffic_averages_the_strict_launches_and_applies_the_corrections(fake_tool, tmp_path):
    nbytes, detail = bench.live_traffic(_args(tmp_path))
    # strict launches of the default arithmetic: scales 1 and 3 -> mean 2
    assert detail["FETCH_SIZE_KB"] == 2000.0 and detail["WRITE_SIZE_KB"] == 200.0 and detail["TCC_EA0_RDREQ"] == 32000.0
    assert nbytes == int(2000.0 * 1024 * 2 + 200.0 * 1024)
    assert detail["fetch_bytes_from_rdreq_x128"] == 32000 * 128
    assert detail["dispatches_averaged"] == {"FETCH_SIZE": 2, "TCC_EA0_RDREQ_sum": 2, "WRITE_SIZE": 2}


def test_live_traffic_gives_up_cleanly(fake_tool, tmp_path, monkeypatch):
    monkeypatch.setenv("FAKE_ROCPROF_FAIL", "1")
    nbytes, detail = bench.live_traffic(_args(tmp_path))
    assert nbytes is None and "exited with 3" in detail["skipped"]
    monkeypatch.delenv("FAKE_ROCPROF_FAIL")
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")                  # this process is being profiled: no nested runs
    nbytes, detail = bench.live_traffic(_args(tmp_path))
    assert nbytes is None and "being profiled" in detail["skipped"]
