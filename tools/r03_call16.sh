#!/bin/bash
# Round-3 GPU call 16: the cold parts of the search kernel out of line (query start, descent, bitmap migration, query end read
# their kernel arguments from the kernarg segment): no spilled scalar registers in the expansion loop.  Suite + configs.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call16
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5
for i in 1 2; do
  echo "== sift1m"
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-boundary 2>/dev/null | python tools/bench_line.py
done
for cfg in glove25 glove25_dot mnist784 random10k; do
  echo "== $cfg"
  timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary 2>/dev/null | python tools/bench_line.py
done
