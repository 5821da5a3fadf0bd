#!/bin/bash
# round 6, call 6: the whole GPU suite on the committed tree (asm chain, micro-diet, hooks read once, strong-scaling plan, mnist784_hbm test), then the driver's command
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06_c6
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r06_c6/suite.log
CFG=sift1m tools/variant_ab.sh r06_c6_sift new10k:10000:X=1 old10k:10000:LIB=lib_r05.so 2>&1 | grep -v "^queries in flight\|^last finishers" | cut -c1-260
timeout 900 python bench.py > gpurun_out/r06_c6/bench.json 2> gpurun_out/r06_c6/bench.log; python tools/bench_line.py < gpurun_out/r06_c6/bench.json; tail -3 gpurun_out/r06_c6/bench.log
