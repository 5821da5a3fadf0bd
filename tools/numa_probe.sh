#!/bin/bash
echo "nodes:"; for n in /sys/devices/system/node/node*; do echo "$(basename $n): $(cat $n/cpulist)"; done
echo "gpu numa:"; for c in /sys/class/drm/card*/device/numa_node; do echo "$c $(cat $c)"; done
nproc
python bench.py --config sift1m --steps 2 --warmup 1 --no-boundary --no-cpu-baseline --no-recall > /dev/null 2>gpurun_out/sweep_build.log
for n in /sys/devices/system/node/node*; do
  cpus=$(cat $n/cpulist)
  echo "== taskset $(basename $n) $cpus"
  taskset -c $cpus python tools/host_call_sweep.py --config sift1m --reps 60 2>&1 | grep -E "^default"
done
echo "== unbound"
python tools/host_call_sweep.py --config sift1m --reps 60 2>&1 | grep -E "^default"
