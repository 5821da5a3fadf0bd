#!/bin/bash
# call 32: the two d = 784 configs recorded again on the tree with the pipelined long rows (bench line + rocprofv3), like `gpu_call.sh record`
cd "$(dirname "$0")/.."
R=r06; O=gpurun_out/${R}_record784; mkdir -p $O
python - > $O/PROVENANCE.json <<PY
import hashlib, json, socket, subprocess, time
def sh(c):
    try: return subprocess.run(c, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception: return ""
h = hashlib.sha256()
for f in ("hnswlib-rs_amd/csrc/search_kernels.inc", "hnswlib-rs_amd/csrc/search_device.hip", "hnswlib-rs_amd/csrc/search_launchers.inc", "bench.py"):
    h.update(open(f, "rb").read())
print(json.dumps({"utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "host": socket.gethostname(),
                  "gpu": sh("rocm-smi --showproductname --csv | tail -n +2 | head -1")[:120], "gpu_unique_id": sh("rocm-smi --showuniqueid --csv | tail -n +2 | head -1")[:80],
                  "rocm": sh("cat /opt/rocm/.info/version"), "source_sha256_16": h.hexdigest()[:16]}))
PY
for cfg in mnist784 mnist784_hbm sift1m; do
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 > $O/bench_$cfg.json 2> $O/bench_$cfg.log
  echo "-- $cfg"; python tools/bench_line.py < $O/bench_$cfg.json | cut -c1-300
done
for cfg in mnist784 mnist784_hbm; do
  timeout 600 tools/profile_round.sh ${R}_record784/prof_$cfg --config $cfg > $O/prof_$cfg.log 2>&1
  python tools/summarize_profile.py $O/prof_$cfg > $O/rocprofv3_summary_$cfg.txt 2>&1
  cp $(find $O/prof_$cfg/kt -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$cfg.csv 2>/dev/null
  echo "-- $cfg"; head -6 $O/rocprofv3_summary_$cfg.txt
  rm -rf $O/prof_$cfg
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
