"""Host-only parallel construction (the reference's parallel_insert path of the product builder) against thread count."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
import hnsw_rs_amd as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
X = bench.synth(n, 128, 0x5EED0001, "clustered")
for nth in (256, 128, 64, 32, 16):
    t0 = time.time()
    hb = H.Hnsw(16, len(X), 16, 200, "DistL2")
    hb.set_build_options(nthreads=nth, fast_arithmetic=False)
    hb.parallel_insert(X)
    print(f"host build, {n} x 128, threads={nth}: {time.time() - t0:.2f} s", flush=True)
    del hb
