#!/bin/bash
# Round-2 GPU call 18: literal pushes as runs with register-held ancestor chains (heap_push_many) -- parity suite, bench, timeline of the literal queries.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out/r02_call18
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --config sift1m --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2>> $O/bench.log | tee -a $O/bench_sift1m.json | python tools/bench_line.py
done
timeout 300 python bench.py --config glove25 --steps 20 --warmup 4 --no-cpu-baseline --no-recall 2>> $O/bench.log | tee -a $O/bench_glove25.json | python tools/bench_line.py
timeout 300 python bench.py --config mnist784 --steps 10 --warmup 2 --no-cpu-baseline --no-recall 2>> $O/bench.log | tee -a $O/bench_mnist784.json | python tools/bench_line.py
HNSW_MI355X_LIB=$ROOT/hnswlib-rs_amd/lib_tl2.so timeout 300 python bench.py --config sift1m --steps 8 --warmup 2 --no-cpu-baseline --no-recall --dump-stats $O/stats_tl2.npy 2>> $O/bench.log | python tools/bench_line.py
