#!/bin/bash
# Round-2 GPU call 1: A/B of the round-1 opt-in experiments against the default build on ONE box (the graph differs per
# box), the N=2 path on a shared device, and a first look at configs 3 and 5 with the shipped kernel.
#   gpurun --timeout 1500 -- 'tools/r02_call1.sh'
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call1
mkdir -p $O
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $*"; }

stamp "default build: GPU suite"
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3

bench() {  # name lib config extra...
  local name=$1 lib=$2 cfg=$3; shift 3
  HNSW_MI355X_LIB=$lib timeout 400 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-recall "$@" \
      --dump-stats $O/stats_${name}_$cfg.npy 2> $O/bench_${name}_$cfg.log | tee $O/bench_${name}_$cfg.json | python tools/bench_line.py
}

stamp "default: sift1m"
bench default "" sift1m
for n in endstate resume valuer specrows; do
  lib=$ROOT/hnswlib-rs_amd/lib_$n.so
  [ -f "$lib" ] || continue
  stamp "$n: tie/parity tests"
  HNSW_MI355X_LIB=$lib timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "strict_ties or search_matches or scheduling or golden or overflow" 2>&1 | tail -3
  stamp "$n: sift1m"
  bench $n $lib sift1m
done

stamp "N=2 on a shared device (gloo): the multi-rank path of bench.py"
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --share-device --backend gloo --nq 5000 --steps 5 --warmup 1 --no-cpu-baseline --no-recall \
    > $O/bench_n2_shared.json 2> $O/bench_n2_shared.log
python tools/bench_line.py < $O/bench_n2_shared.json || tail -5 $O/bench_n2_shared.log

stamp "default: mnist784"
bench default "" mnist784
stamp "default: glove25"
bench default "" glove25
if [ -f hnswlib-rs_amd/lib_cosg.so ]; then
  lib=$ROOT/hnswlib-rs_amd/lib_cosg.so
  stamp "cosg: cosine parity tests"
  HNSW_MI355X_LIB=$lib timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "Cosine or golden or distances" 2>&1 | tail -3
  stamp "cosg: glove25"
  bench cosg $lib glove25
fi
stamp done
