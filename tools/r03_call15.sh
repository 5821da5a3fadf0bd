#!/bin/bash
# Round-3 GPU call 15: soak at full size with the final kernels: fresh batches (clustered and uniform alternating) of every config
# against the oracle on the same dump.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for cfg in sift1m glove25_dot glove25 mnist784; do
  timeout 200 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-concurrent --no-boundary > /dev/null 2>&1
  echo "== $cfg"
  timeout 600 python tools/soak_parity.py --config $cfg --batches 6 2>&1 | tail -2
done
