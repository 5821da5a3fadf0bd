#!/bin/bash
# call 28: heap_pop3 with its first round aligned to the LDS part of the heap: lane lab + filtered tests, then A/B (variant lib_pop5 = rounds as before)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call28; mkdir -p $O
P=$PWD/hnswlib-rs_amd
timeout 1500 python -m pytest tests/test_gpu_lane_lab.py -m gpu -x -q 2>&1 | tail -4
timeout 1500 python -m pytest tests -m gpu -x -q -k "filter or literal or exact or tie" 2>&1 | tail -4
run() { tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 6 --warmup 2 --no-recall --no-cpu-baseline --no-traffic --no-concurrent > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
f=j["boundary"]["filtered"]
out=[]
for nqk,v in f.items():
    if not isinstance(v,dict): continue
    for sel,w in v.items():
        if isinstance(w,dict) and "queries_per_s" in w: out.append(f"{nqk[:6]} {sel}: {w['queries_per_s']/1e3:.1f}k q/s, {w['per_query']['us_per_expansion_p50']} us/exp")
print(sys.argv[2], "value", j["value"], "kernel_ms", j["roofline"]["kernel_ms"], "|", " | ".join(out))
PY
}
run aligned X=1
run rounds5 HNSW_MI355X_LIB=$P/lib_pop5.so
run aligned2 X=1
run rounds5b HNSW_MI355X_LIB=$P/lib_pop5.so
