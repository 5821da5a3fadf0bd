#!/bin/bash
# Round-3 GPU call 9: literal candidate heap replayed in runs of pushes; single-copy answers + pipelined gather in the host-buffer
# entry points.  Full GPU suite, then the configs and the boundary timings.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call9
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -8
for cfg in sift1m glove25 glove25_dot mnist784; do
  echo "== $cfg"
  timeout 400 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-concurrent --no-boundary 2>/dev/null | python tools/bench_line.py
done
echo "== boundary timings (sift1m)"
for i in 1 2; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent 2>$O/boundary.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): j=json.loads(l); print(j['value'], json.dumps(j['boundary']))"; done
