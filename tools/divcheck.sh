#!/bin/bash
# usage: divcheck.sh  -> prints whether the strict S=1 main loop is divergent
cd /root/repo
tools/kasm.sh 0 0 /tmp/dc.s
awk '/^_ZN7hnswgpu12_GLOBAL__N_118hnsw_search_kernelILi0ELi1ELi0ELb1EEEv/{f=1} f{print} /s_endpgm/{if(f){exit}}' /tmp/dc.s > /tmp/dc_b.s
python3 - <<'PY'
lines=open('/tmp/dc_b.s').read().split('\n')
hdrs=[i for i,l in enumerate(lines) if 'This Loop Header: Depth=2' in l]
for h in hdrs:
    pre=[l for l in lines[max(0,h-8):h] if 's_andn2_b64 exec, exec' in l]
    print('header at', h, 'divergent' if pre else 'uniform')
PY
