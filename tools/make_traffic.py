#!/usr/bin/env python3
"""profiles/traffic.json from the committed rocprofv3 summaries (tools/summarize_profile.py output) and bench lines:
per config, fabric-side bytes per launch of hnsw_search_kernel = FETCH_SIZE (KB) x 1024 x 2 (gfx950 correction,
MI355X_MICROARCH.md HBM section; cross-check TCC_EA0_RDREQ x 128 B) + WRITE_SIZE (KB) x 1024, from separate --pmc
passes, next to the algorithmic bytes of the same workload.  usage: tools/make_traffic.py [round tag, default r02]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
# provenance of the record the counters come from (tools/gpu_call.sh record writes it on the GPU box; the commit is added here,
# on the machine that has the history): the bench line quotes it, so that a reader sees how stale the counters are
prov = {}
pp = os.path.join(ROOT, "profiles", f"{tag}_PROVENANCE.json")
if os.path.exists(pp):
    prov = json.load(open(pp))
if "commit" not in prov:
    try:
        import subprocess
        prov["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        prov["commit_note"] = "HEAD when profiles/traffic.json was generated; source_sha256_16 identifies the kernel sources the box ran"
    except Exception:
        pass
out = {}
for cfg in ("sift1m", "glove25", "glove25_dot", "mnist784", "mnist784_hbm"):
    sp = os.path.join(ROOT, "profiles", f"{tag}_{cfg}_rocprofv3_summary.txt")
    bp = os.path.join(ROOT, "profiles", f"{tag}_bench_{cfg}.json")
    if not (os.path.exists(sp) and os.path.exists(bp)):
        continue
    bench = json.loads([l for l in open(bp) if l.startswith("{")][-1])
    vals = {"strict": {}, "lean": {}}
    for line in open(sp):
        m = re.match(r"\s+hnsw_search_kernel<[^>]*, (true|false)>\s+(\S+)\s+avg\s+(\S+)", line)
        if m:
            vals["strict" if m.group(1) == "true" else "lean"][m.group(2)] = float(m.group(3))
    ent = {}
    for mode, v in vals.items():
        if "FETCH_SIZE" not in v:
            continue
        ent[mode] = {"FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v.get("WRITE_SIZE"), "TCC_EA0_RDREQ": v.get("TCC_EA0_RDREQ_sum"),
                     "hbm_bytes_per_launch": int(v["FETCH_SIZE"] * 1024 * 2 + v.get("WRITE_SIZE", 0.0) * 1024),
                     "fetch_bytes_from_rdreq_x128": int(v.get("TCC_EA0_RDREQ_sum", 0.0) * 128)}
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    ent["algorithmic_bytes_per_launch"] = alg
    if "strict" in ent:
        ent["traffic_over_algorithmic_strict"] = round(ent["strict"]["hbm_bytes_per_launch"] / alg, 3)
    ent["workload"] = bench["config"]["workload"]
    ent["data"] = "clustered" if "clustered" in bench["data"] else "uniform"
    ent["source"] = (f"profiles/{tag}_{cfg}_rocprofv3_summary.txt: FETCH_SIZE (KB) x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md HBM "
                     f"section; cross-check TCC_EA0_RDREQ x 128 B) + WRITE_SIZE (KB) x 1024, separate --pmc passes of `python bench.py "
                     f"--config {cfg} --steps 5 --warmup 1`; fabric-side requests: Infinity-Cache hits are included")
    ent["provenance"] = dict(prov, round=tag)
    out[cfg] = ent
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for cfg, ent in out.items():
    print(cfg, ent.get("traffic_over_algorithmic_strict"), ent["strict"]["hbm_bytes_per_launch"] if "strict" in ent else None)
