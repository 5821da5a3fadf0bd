#!/bin/bash
# Round-2 GPU call 14: the probability distances (Hellinger / Jeffreys / JensenShannon) on the device; full suite.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call14
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_golden.py -m gpu -x -q -k "probability or golden" 2>&1 | tail -8
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2> $O/bench.log | tee $O/bench_sift1m.json | python tools/bench_line.py
