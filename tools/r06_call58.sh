#!/bin/bash
# call 58: one rank's share of config 4 as the 8-GPU run executes it: 12 500 queries per step, the packed all-gather over RCCL (one-rank group) in the timed region
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call58; mkdir -p $O
timeout 900 python bench.py --gpus 1 --exchange --backend nccl --nq 12500 --steps 20 --warmup 5 --no-cpu-baseline --no-boundary --no-traffic > $O/bench_per_rank.json 2> $O/bench_per_rank.log
python - <<PY
import json
j=json.load(open("$O/bench_per_rank.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], j.get("gather_ms"), j.get("rccl"))
PY
