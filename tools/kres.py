#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one kernel translation unit (cross-compiles, no GPU needed):
    tools/kres.py METRIC PART [extra -D flags ...]      e.g.  tools/kres.py 0 0"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
metric, part, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "--offload-arch=gfx950",
       "-fhip-fp32-correctly-rounded-divide-sqrt", *extra, f"-DHNSW_THIS_METRIC={metric}", f"-DHNSW_PART={part}",
       "-Rpass-analysis=kernel-resource-usage", "-c", "search_kernels_tu.hip", "-o", f"/tmp/kres_{metric}_{part}.o"]
p = subprocess.run(cmd, cwd=os.path.join(root, "hnswlib-rs_amd", "csrc"), capture_output=True, text=True)
rows, cur = [], {}
for line in p.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split()[0]] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"hnswgpu::\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    print(f"{name:58s} vgpr {r.get('VGPRs', '?'):>4s} sgpr {r.get('TotalSGPRs', '?'):>4s} scratch {r.get('ScratchSize', '?'):>4s} occ {r.get('Occupancy', '?')}")
sys.exit(p.returncode)
