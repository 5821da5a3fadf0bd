#!/bin/bash
# call 40: the whole GPU suite, smoke() and the driver's bench command on the tree with hnsw_search_pair_kernel in it (off by default)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call40; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.log; python tools/bench_line.py < $O/bench_driver.json | cut -c1-400
