#!/bin/bash
# Settings and variant libraries side by side on one box:  [CFG=glove25] tools/variant_ab.sh TAG SPEC...   with SPEC = name:nq:ENV=V,ENV=V
# (variant libraries from tools/mkvariant.sh: LIB=lib_x.so).  One line per setting + how the launch ended (tools/tail_report.py).
TAG=${1:-crit_ab}; shift; O=gpurun_out/$TAG; mkdir -p $O
P=$PWD/hnswlib-rs_amd
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; nq=${rest%%:*}; envs=${rest#*:}
  [ "$envs" = "$rest" ] && envs=X=1
  envs=$(echo "$envs" | tr ',' ' ' | sed "s#LIB=#HNSW_MI355X_LIB=$P/#")
  env $envs timeout 300 python bench.py --config ${CFG:-sift1m} --nq $nq --steps 12 --warmup 3 --no-recall --no-boundary --no-cpu-baseline --no-traffic \
      --dump-stats $O/st_$name.npy 2> $O/err_$name.log > $O/bench_$name.json
  echo "== $name nq=$nq $envs"
  python tools/bench_line.py < $O/bench_$name.json | cut -c1-150
  python tools/tail_report.py $O/st_$name.npy 2>&1 | head -4 | cut -c1-330
done
