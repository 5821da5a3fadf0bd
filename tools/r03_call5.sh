#!/bin/bash
# Round-3 GPU call 5: speculative id-row prefetch issued AFTER the ids arrived (it was waited for at once), wave-level LDS fences
# instead of workgroup barriers in the expansion, log stores deferred behind the id wait; pinned + parallel staging of the
# host-buffer entry points and the one-slab FFI answer.  Full GPU suite first.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call5
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -8
for v in nors default nors default; do
  echo "== sift1m $v"
  if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-boundary 2>/dev/null | python tools/bench_line.py
done
for cfg in glove25 glove25_dot mnist784 random10k; do
  for v in nors default; do
    echo "== $cfg $v"
    if [ $v = default ]; then unset HNSW_MI355X_LIB; else export HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_$v.so; fi
    timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary 2>/dev/null | python tools/bench_line.py
  done
done
unset HNSW_MI355X_LIB
echo "== sift1m default: parity vs oracle + boundary"
timeout 600 python bench.py --steps 20 --warmup 4 > $O/bench_sift1m.json 2> $O/bench_sift1m.log; python tools/bench_line.py < $O/bench_sift1m.json
python -c "
import json
j=json.load(open('$O/bench_sift1m.json')); print(json.dumps(j['boundary'])); print(j['recall_at_10'], j['cpu_baseline']['value'])"
