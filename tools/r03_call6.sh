#!/bin/bash
# Round-3 GPU call 6: select_neighbours on the device (hnsw_build_select_kernel): window-1 builds against the oracle's serial graph,
# construction tests, build time of 1M x 128; boundary timings (pinned staging, one-slab FFI answers).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call6
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "construction or window_1 or reloaded or keeps_growing" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -8
echo "== build timing (1M x 128, fresh cache), device select"
HNSWGPU_BUILD_TIMING=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-concurrent --no-boundary --cache-dir /tmp/fresh_cache 2> $O/build.log | python tools/bench_line.py
grep -E "hnswgpu build|built in" $O/build.log
python -c "
import json,sys
" 
echo "== the same with select on the host"
HNSWGPU_HOST_SELECT=1 HNSWGPU_BUILD_TIMING=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-concurrent --no-boundary --cache-dir /tmp/fresh_cache2 2> $O/build_host.log | python tools/bench_line.py
grep -E "hnswgpu build|built in" $O/build_host.log
echo "== recall of both graphs"
for c in /tmp/fresh_cache /tmp/fresh_cache2; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-boundary --cache-dir $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): j=json.loads(l); print(j['recall_at_10'], j['value'])"; done
echo "== boundary timings (sift1m)"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent 2>$O/boundary.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): j=json.loads(l); print(j['value'], json.dumps(j['boundary']))"
