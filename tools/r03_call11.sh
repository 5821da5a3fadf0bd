#!/bin/bash
# Round-3 GPU call 11: boundary timings after the staging fix; literal candidate heap with more of its top in LDS; RCCL's own
# words on two ranks sharing a device.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call11
mkdir -p $O
for c in 256 512 1024; do
  for cfg in sift1m glove25_dot; do
    echo "== $cfg HNSWGPU_CAND_LDS=$c"
    HNSWGPU_CAND_LDS=$c timeout 300 python bench.py --config $cfg --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-boundary --no-concurrent 2>/dev/null | python tools/bench_line.py
  done
done
echo "== boundary timings (sift1m)"
for i in 1 2 3; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-concurrent 2>$O/boundary.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): j=json.loads(l); b=j['boundary']; print(j['value'], b['host_buffers_queries_per_s'], b['ffi_parallel_search_neighbours_f32_queries_per_s'])"; done
echo "== N = 2 plain command"
timeout 400 python bench.py --gpus 2 --share-device --backend nccl --nq 5000 --steps 10 --warmup 2 --no-cpu-baseline --no-recall > $O/n2.json 2> $O/n2.log
grep "rccl probe" $O/n2.log | tail -12 | cut -c1-260
python -c "
import json
j=[json.loads(l) for l in open('$O/n2.json') if l.startswith('{')][-1]
print(j['value'], j['n_gpus'], j['rccl'])"
