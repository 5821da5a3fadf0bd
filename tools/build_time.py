"""GPU-assisted construction of the config-2 graph under different host settings (HNSWGPU_BUILD_TIMING=1 prints the split)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
import hnsw_rs_amd as H
X = bench.synth(1_000_000, 128, 0x5EED0001, "clustered")
for fast, nth in ((False, 32), (False, 64), (False, 128), (False, 256), (True, 64)):
    t0 = time.time()
    hb = H.Hnsw(16, len(X), 16, 200, "DistL2")
    hb.set_build_options(nthreads=nth, gpu_device=0, gpu_window=0, fast_arithmetic=fast)
    hb.parallel_insert(X)
    print(f"fast_arithmetic={fast} threads={nth}: built in {time.time() - t0:.2f} s", flush=True)
    del hb
