#!/bin/bash
# Round-3 record: full GPU suite, bench lines of every BASELINE config (+ config 3's DistDot twin), the N = 2 plain command,
# rocprofv3 kernel trace + PMC passes per config, the literal-heap kernel under filtered search.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_final
mkdir -p $O
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $*"; }
stamp "GPU suite"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
stamp "bench lines (CPU baseline, recall, parity at full size, boundary timings)"
for cfg in sift1m glove25 glove25_dot mnist784 random10k; do
  timeout 500 python bench.py --config $cfg --steps 20 --warmup 4 > $O/bench_$cfg.json 2> $O/bench_$cfg.log
  echo "-- $cfg"; python tools/bench_line.py < $O/bench_$cfg.json
  grep -E "built in" $O/bench_$cfg.log
done
stamp "N = 2 as a plain command (both ranks on the one device)"
timeout 400 python bench.py --gpus 2 --share-device --backend nccl --nq 5000 --steps 10 --warmup 2 --no-cpu-baseline --no-recall \
    2> $O/bench_sift1m_n2_shared_device.log | grep '^{' > $O/bench_sift1m_n2_shared_device.json
python -c "
import json
j=[json.loads(l) for l in open('$O/bench_sift1m_n2_shared_device.json') if l.startswith('{')][-1]
print(j['value'], j['n_gpus'], j['rccl'])" || tail -5 $O/bench_sift1m_n2_shared_device.log
stamp "rocprofv3 per config"
for cfg in sift1m glove25 glove25_dot mnist784; do
  timeout 600 tools/profile_round.sh r03_final/prof_$cfg --config $cfg > $O/prof_$cfg.log 2>&1
  python tools/summarize_profile.py $O/prof_$cfg > $O/rocprofv3_summary_$cfg.txt 2>&1
  cp $(find $O/prof_$cfg/kt -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$cfg.csv 2>/dev/null
  echo "-- $cfg"; head -10 $O/rocprofv3_summary_$cfg.txt
  rm -rf $O/prof_$cfg
done
stamp "filtered search (hnsw_search_exact_kernel) under rocprofv3"
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$O/prof_filtered --output-format csv -- python $ROOT/tools/filtered_run.py --config sift1m > $ROOT/$O/filtered.log 2>&1)
tail -2 $O/filtered.log
python tools/summarize_profile.py $O/prof_filtered > $O/rocprofv3_summary_filtered_sift1m.txt 2>&1
find $O/prof_filtered -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'python3 - {} <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5: print(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, "us avg")
PY'
cp $(find $O/prof_filtered -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_filtered_sift1m.csv 2>/dev/null
rm -rf $O/prof_filtered
stamp done
