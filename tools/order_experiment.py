"""Experiment (not part of the product): how much of the batch time is scheduling?  Runs the lean search kernel on
the bench workload with the queries (a) in input order, (b) sorted by their measured expansion count (an oracle
longest-first order), (c) sorted by a noisy version of it (rank correlation ~0.75, what a cheap predictor gives)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hnsw_rs_amd as H

cfg = bench.CONFIGS["sift1m"]
n, d, k, ef, nq = cfg["n"], cfg["d"], cfg["k"], cfg["ef"], cfg["nq"]
H.build_native(); lib = H.lib()
X = bench.synth(n, d, 0x5EED0001, "clustered")
hb = H.Hnsw(cfg["M"], n, 16, cfg["efc"], cfg["dist"]); hb.set_build_options(nthreads=0, fast_arithmetic=True)
t0 = time.time(); hb.parallel_insert(X); print("built", time.time() - t0, flush=True)
hb.upload(0); hb.set_strict_ties(False)
dev = torch.device("cuda", 0)
Q = bench.synth(nq, d, 0x5EED0002, "clustered")
out_ids = torch.zeros((nq, k), dtype=torch.int64, device=dev); out_d = torch.zeros((nq, k), dtype=torch.float32, device=dev)
out_l = torch.zeros((nq, k), dtype=torch.uint8, device=dev); out_r = torch.zeros((nq, k), dtype=torch.int32, device=dev)
out_c = torch.zeros((nq,), dtype=torch.int32, device=dev); stats = torch.zeros((nq, 8), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
def run(Qh, reps=8):
    Qd = torch.from_numpy(np.ascontiguousarray(Qh)).to(dev)
    ms = []
    for i in range(reps + 2):
        rc = lib.hnswgpu_search_batch_device(hb.handle, Qd.data_ptr(), nq, d, k, ef, out_ids.data_ptr(), out_d.data_ptr(),
                                             out_l.data_ptr(), out_r.data_ptr(), out_c.data_ptr(), stats.data_ptr(), stream.cuda_stream)
        assert rc == 0
        torch.cuda.synchronize(dev)
        if i >= 2: ms.append(hb.last_search_kernel_ms())
    return float(np.mean(ms)), stats.cpu().numpy().astype(np.int64)
ms0, st = run(Q)
nexp = st[:, 1].astype(np.float64)
print("input order ms", ms0)
order = np.argsort(-nexp); ms1, _ = run(Q[order]); print("oracle longest-first ms", ms1)
order = np.argsort(nexp); ms1b, _ = run(Q[order]); print("oracle shortest-first ms", ms1b)
rng = np.random.default_rng(1)
from scipy.stats import spearmanr
for sigma in (10.0, 20.0, 40.0):
    noisy = nexp + rng.normal(0, sigma, nq)
    rho = spearmanr(noisy, nexp).statistic
    ms2, _ = run(Q[np.argsort(-noisy)]); print(f"noisy (sigma {sigma}, spearman {rho:.2f}) longest-first ms", ms2)
