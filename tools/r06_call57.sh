#!/bin/bash
# call 57: last sanity of the in-tree library (rebuilt from the recorded sources): smoke(), the parity and pair tests, the driver's command without its extras
cd "$(dirname "$0")/.."
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pair_search.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-boundary --no-traffic --no-concurrent | python tools/bench_line.py | cut -c1-260
