#!/bin/bash
# ONE entry point for everything that is run on the GPU box:   gpurun --timeout S -- tools/gpu_call.sh <experiment> [args]
# Every experiment writes under gpurun_out/<tag>/ and appends one line to gpurun_out/<tag>/CALL.log; what is worth keeping is
# copied into profiles/ by hand and recorded in tools/GPU_CALLS.md (the log of the round's calls).
# Graphs differ per box (GPU-assisted construction is racy by design), so only numbers of ONE call compare.
#   suite [pytest args]          the GPU test suite (default: all of it)
#   bench TAG [bench args]       one bench.py line -> gpurun_out/TAG/bench.json (+ .log), summary on stdout
#   ab TAG CFG ENV=V[,ENV=V] ... the same short bench under several environments (knobs, variant libraries), base first and last
#   profile TAG [bench args]     rocprofv3 kernel trace + the PMC passes (tools/profile_round.sh) + summary
#   record ROUND                 the round's record: suite, bench lines of every config, N=2 plain command, profiles
#   sh 'command'                 anything else, verbatim
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $*"; }
line() { python tools/bench_line.py; }
EXP=${1:-}; shift || true
case "$EXP" in
  suite)
    stamp "GPU suite $*"
    timeout 1500 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15
    ;;
  bench)
    TAG=$1; shift
    mkdir -p gpurun_out/$TAG
    stamp "bench $*"
    timeout 900 python bench.py "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.log
    line < gpurun_out/$TAG/bench.json
    grep -E "built in|hnswgpu (launch|host call)" gpurun_out/$TAG/bench.log | sort | uniq -c | sort -rn | head -8
    ;;
  ab)
    TAG=$1; CFG=$2; shift 2
    mkdir -p gpurun_out/$TAG
    run() {
      local tag=$1; shift
      echo "== $CFG $tag"
      env HNSWGPU_TRACE_LAUNCH=1 "$@" timeout 400 python bench.py --config $CFG --steps 12 --warmup 3 --no-recall --no-boundary --no-cpu-baseline --no-traffic \
          --dump-stats gpurun_out/$TAG/st_$tag.npy 2> gpurun_out/$TAG/err_$tag.log | tee gpurun_out/$TAG/bench_$tag.json | line | cut -c1-220
      grep "hnswgpu launch" gpurun_out/$TAG/err_$tag.log | sort | uniq -c | sort -rn | head -2
    }
    run base X=1
    i=0
    for spec in "$@"; do i=$((i+1)); run v$i $(echo "$spec" | tr ',' ' '); done
    run base_again X=1
    ;;
  profile)
    TAG=$1; shift
    stamp "rocprofv3 $*"
    timeout 900 tools/profile_round.sh $TAG/prof "$@" > /dev/null 2>&1
    python tools/summarize_profile.py gpurun_out/$TAG/prof > gpurun_out/$TAG/rocprofv3_summary.txt 2>&1
    cp $(find gpurun_out/$TAG/prof/kt -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/rocprofv3_kernel_stats.csv 2>/dev/null
    head -40 gpurun_out/$TAG/rocprofv3_summary.txt
    rm -rf gpurun_out/$TAG/prof
    ;;
  record)
    R=$1; O=gpurun_out/${R}_record
    mkdir -p $O
    # where and on what tree the record was taken (profiles/traffic.json and the bench line quote it)
    python - > $O/PROVENANCE.json <<PY
import hashlib, json, socket, subprocess, time
def sh(c):
    try: return subprocess.run(c, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception: return ""
h = hashlib.sha256()
for f in ("hnswlib-rs_amd/csrc/search_kernels.inc", "hnswlib-rs_amd/csrc/search_pair.inc", "hnswlib-rs_amd/csrc/search_device.hip", "hnswlib-rs_amd/csrc/search_launchers.inc", "bench.py"):
    h.update(open(f, "rb").read())
print(json.dumps({"utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "host": socket.gethostname(),
                  "gpu": sh("rocm-smi --showproductname --csv | tail -n +2 | head -1")[:120], "gpu_unique_id": sh("rocm-smi --showuniqueid --csv | tail -n +2 | head -1")[:80],
                  "rocm": sh("cat /opt/rocm/.info/version"), "source_sha256_16": h.hexdigest()[:16]}))
PY
    stamp "GPU suite"
    timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
    stamp "bench lines (CPU baseline, recall, parity at full size, boundary timings)"
    for cfg in sift1m glove25 glove25_dot mnist784 mnist784_hbm random10k; do
      timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 > $O/bench_$cfg.json 2> $O/bench_$cfg.log
      echo "-- $cfg"; line < $O/bench_$cfg.json
      grep -E "built in" $O/bench_$cfg.log
    done
    stamp "config 4's batch on one GPU (100 000 queries per call)"
    timeout 600 python bench.py --config sift1m --nq 100000 --steps 5 --warmup 2 --no-boundary --no-traffic > $O/bench_sift1m_nq100k.json 2> $O/bench_sift1m_nq100k.log
    line < $O/bench_sift1m_nq100k.json
    stamp "N = 2 as a plain command (both ranks on the one device)"
    timeout 600 python bench.py --gpus 2 --share-device --backend nccl --steps 5 --warmup 2 --no-cpu-baseline --no-recall \
        2> $O/bench_sift1m_n2_shared_device.log | grep '^{' > $O/bench_sift1m_n2_shared_device.json
    python -c "
import json
j=[json.loads(l) for l in open('$O/bench_sift1m_n2_shared_device.json') if l.startswith('{')][-1]
print(j['value'], j['n_gpus'], j['scaling'], j['config']['queries_total'], j['one_gpu_same_batch_queries_per_s'], j['gather_ms'], j['rccl'])" || tail -5 $O/bench_sift1m_n2_shared_device.log
    stamp "rocprofv3 per config"
    for cfg in sift1m glove25 glove25_dot mnist784 mnist784_hbm; do
      timeout 600 tools/profile_round.sh ${R}_record/prof_$cfg --config $cfg > $O/prof_$cfg.log 2>&1
      python tools/summarize_profile.py $O/prof_$cfg > $O/rocprofv3_summary_$cfg.txt 2>&1
      cp $(find $O/prof_$cfg/kt -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$cfg.csv 2>/dev/null
      echo "-- $cfg"; head -12 $O/rocprofv3_summary_$cfg.txt
      rm -rf $O/prof_$cfg
    done
    ;;
  sh)
    stamp "$*"
    bash -c "$*"
    ;;
  *)
    echo "usage: tools/gpu_call.sh suite|bench|ab|profile|record|sh ..."; exit 2 ;;
esac
stamp done
