#!/bin/bash
# Is the host-buffer boundary sensitive to the socket the caller runs on?  tools/host_call_sweep.py under `taskset` for every NUMA
# node of the box, then unbound (GPU_CALLS.md call 30: it is not).  usage (on the GPU box): tools/numa_probe.sh
echo "nodes:"; for n in /sys/devices/system/node/node*; do echo "$(basename $n): $(cat $n/cpulist)"; done
echo "gpu numa:"; for c in /sys/class/drm/card*/device/numa_node; do echo "$c $(cat $c)"; done
nproc
python bench.py --config sift1m --steps 2 --warmup 1 --no-boundary --no-cpu-baseline --no-recall --no-traffic > /dev/null 2>gpurun_out/sweep_build.log
for n in /sys/devices/system/node/node*; do
  cpus=$(cat $n/cpulist)
  echo "== taskset $(basename $n) $cpus"
  taskset -c $cpus python tools/host_call_sweep.py --config sift1m --reps 60 2>&1 | grep -E "^default"
done
echo "== unbound"
python tools/host_call_sweep.py --config sift1m --reps 60 2>&1 | grep -E "^default"
