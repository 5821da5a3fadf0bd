#!/bin/bash
# call 39: config 3 / 3' strict: resident workgroups per CU against the three lock-step rounds of 10 000 equal-length searches
cd "$(dirname "$0")/.."
for cfg in glove25 glove25_dot; do
CFG=$cfg tools/variant_ab.sh r06_call39_$cfg w16:10000 w15:10000:HNSWGPU_STRICT_WG_PER_CU=15 w14:10000:HNSWGPU_STRICT_WG_PER_CU=14 w12:10000:HNSWGPU_STRICT_WG_PER_CU=12 2>&1 | grep -v "^$" | grep -v "last finishers" | grep -v "^first round" | cut -c1-330
done
