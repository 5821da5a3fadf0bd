#!/bin/bash
# Round-2 GPU call 8: the round's record -- full GPU suite, bench lines of every BASELINE config, rocprofv3 kernel trace
# + PMC passes per config, the N = 2 path on the shared device.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_final
mkdir -p $O
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $*"; }
stamp "GPU suite"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
stamp "bench lines (CPU baseline, recall, parity at full size)"
for cfg in sift1m glove25 mnist784 random10k; do
  timeout 500 python bench.py --config $cfg --steps 20 --warmup 4 > $O/bench_$cfg.json 2> $O/bench_$cfg.log
  echo "-- $cfg"; python tools/bench_line.py < $O/bench_$cfg.json
done
stamp "N = 2 on the shared device (gloo)"
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 2 --share-device --backend gloo --nq 5000 --steps 10 --warmup 2 --no-cpu-baseline \
    > $O/bench_sift1m_n2_shared_device.json 2> $O/bench_sift1m_n2_shared_device.log
python tools/bench_line.py < $O/bench_sift1m_n2_shared_device.json || tail -5 $O/bench_sift1m_n2_shared_device.log
stamp "rocprofv3 per config"
for cfg in sift1m glove25 mnist784; do
  timeout 600 tools/profile_round.sh r02_final/prof_$cfg --config $cfg > $O/prof_$cfg.log 2>&1
  python tools/summarize_profile.py $O/prof_$cfg > $O/rocprofv3_summary_$cfg.txt 2>&1
  cp $(find $O/prof_$cfg/kt -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$cfg.csv 2>/dev/null
  echo "-- $cfg"; head -12 $O/rocprofv3_summary_$cfg.txt
  rm -rf $O/prof_$cfg
done
stamp done
