#!/bin/bash
# Round-3 GPU call 8: worker pool + pinned patch buffer in the GPU-assisted build; pool-staged host-buffer / FFI searches.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call8
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "construction or window_1 or reloaded or keeps_growing or capi or symbols or sharded or concurrent or begin_end" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -8
echo "== build timing (1M x 128, fresh cache)"
HNSWGPU_BUILD_TIMING=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-concurrent --no-boundary --cache-dir /tmp/fresh_cache 2> $O/build.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): j=json.loads(l); print(j['recall_at_10'], j['value'], j['setup_s'])"
grep -E "hnswgpu build|built in" $O/build.log
echo "== build timing glove25 (1.2M x 25 cosine M=24 efc=400)"
HNSWGPU_BUILD_TIMING=1 timeout 400 python bench.py --config glove25 --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-boundary --cache-dir /tmp/fresh_cache 2> $O/build_glove.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): j=json.loads(l); print(j['recall_at_10'], j['value'], j['setup_s'])"
grep -E "hnswgpu build|built in" $O/build_glove.log
echo "== boundary timings (sift1m)"
for i in 1 2; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent 2>$O/boundary.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): j=json.loads(l); print(j['value'], json.dumps(j['boundary']))"; done
