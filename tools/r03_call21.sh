#!/bin/bash
# Round-3 GPU call 21: the tree at the end of the round -- full GPU suite, the driver's bench command.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call21
mkdir -p $O
T0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? [$(( $(date +%s) - T0 )) s]"; tail -3 $O/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log; echo "bench rc=$? [$(( $(date +%s) - T0 )) s]"
python tools/bench_line.py < $O/bench.json | cut -c1-400
python -c "
import json
j=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print(j['boundary']); print(j['cpu_baseline'])"
