#!/bin/bash
# call 25: the literal (filtered) kernel at more resident workgroups: launch bounds x LDS per workgroup
cd "$(dirname "$0")/.."
O=gpurun_out/r06_call25; mkdir -p $O
P=$PWD/hnswlib-rs_amd
run() { tag=$1; shift
  env HNSWGPU_TRACE_LAUNCH=1 "$@" timeout 600 python bench.py --steps 2 --warmup 1 --no-recall --no-cpu-baseline --no-traffic --no-concurrent > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
f=j["boundary"]["filtered"]
out=[]
for nqk,v in f.items():
    if not isinstance(v,dict): continue
    for sel,w in v.items():
        if isinstance(w,dict) and "queries_per_s" in w: out.append(f"{nqk[:6]} {sel}: {w['queries_per_s']/1e3:.1f}k q/s, {w['per_query']['us_per_expansion_p50']} us/exp")
print(sys.argv[2], "|", " | ".join(out))
PY
  grep "literal kernel" $O/$tag.err | sort | uniq -c | sort -rn | head -2 | cut -c1-200
}
run lb4 X=1
run lb4_lds8k HNSWGPU_EXACT_LDS=8192
run lb5 HNSW_MI355X_LIB=$P/lib_lbx5.so HNSWGPU_EXACT_LDS=8192
run lb6 HNSW_MI355X_LIB=$P/lib_lbx6.so HNSWGPU_EXACT_LDS=6656
run lb8 HNSW_MI355X_LIB=$P/lib_lbx8.so HNSWGPU_EXACT_LDS=5120
run lb4_again X=1
