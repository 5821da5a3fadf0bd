#!/bin/bash
# call 26: the literal heaps without waiting for their global stores (same-wave, same-address order is the hardware's)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call26; mkdir -p $O
P=$PWD/hnswlib-rs_amd
HNSW_MI355X_LIB=$P/lib_hord.so timeout 1200 python -m pytest tests -m gpu -x -q -k "filter or literal or exact or heap or tie" 2>&1 | tail -4
run() { tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --no-recall --no-cpu-baseline --no-traffic --no-concurrent > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
f=j["boundary"]["filtered"]
out=[]
for nqk,v in f.items():
    if not isinstance(v,dict): continue
    for sel,w in v.items():
        if isinstance(w,dict) and "queries_per_s" in w: out.append(f"{nqk[:6]} {sel}: {w['queries_per_s']/1e3:.1f}k q/s, {w['per_query']['us_per_expansion_p50']} us/exp parity {w.get('parity_vs_oracle',{}).get('ids_distance_bits_counts_identical')}")
print(sys.argv[2], "value", j["value"], "|", " | ".join(out))
PY
}
run base X=1
run ordered HNSW_MI355X_LIB=$P/lib_hord.so
run base2 X=1
run ordered2 HNSW_MI355X_LIB=$P/lib_hord.so
