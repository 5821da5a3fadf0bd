#!/bin/bash
# Round-3 GPU call 14: the final tree: full GPU suite, smoke, the driver's bench command, a 1M build.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r03_call14
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== driver command"
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log ) 2>&1 | grep real
python tools/bench_line.py < $O/bench.json
grep "built in" $O/bench.log
python -c "
import json
j=[json.loads(l) for l in open('$O/bench.json') if l.startswith('{')][-1]
print(json.dumps(j['boundary'])); print(j['roofline']['kernel_ms_by_batch_min_median_max'])"
