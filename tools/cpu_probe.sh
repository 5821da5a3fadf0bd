#!/bin/bash
# what the GPU box's host side really offers: cgroup CPU quota, visible CPUs, and how a pure-compute load scales with threads
echo "nproc: $(nproc)   cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
echo "cfs (v1): $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
lscpu | grep -E "Model name|Socket|Core\(s\)|Thread\(s\)|NUMA node|MHz" | head -12
cat > /tmp/spin.c <<'C'
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static volatile int go = 0;
static void* work(void* p) { double* out = p; while (!go) ; double x = 1.0; struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0); long it = 0;
  do { for (int i = 0; i < 1000000; ++i) x = x * 1.0000001 + 1e-9; it += 1000000; clock_gettime(CLOCK_MONOTONIC, &t1); } while ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) < 1.0);
  out[0] = it; out[1] = x; return 0; }
int main(int c, char** v) { int n = atoi(v[1]); pthread_t* t = malloc(n * sizeof *t); double* o = calloc(2 * n, sizeof *o);
  for (int i = 0; i < n; ++i) pthread_create(&t[i], 0, work, o + 2 * i); go = 1; double s = 0; for (int i = 0; i < n; ++i) { pthread_join(t[i], 0); s += o[2 * i]; }
  printf("%3d threads: %.2f G iterations/s in all, %.3f per thread\n", n, s / 1e9, s / 1e9 / n); return 0; }
C
gcc -O1 -pthread /tmp/spin.c -o /tmp/spin && for n in 1 8 16 32 64 128 256; do /tmp/spin $n; done
