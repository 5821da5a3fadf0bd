import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import hnsw_rs_amd as native
import oracle_lib as oracle
from conftest import uniform
import tempfile, pathlib
from test_gpu_round6 import _device_call_with_stats
from test_gpu_parity import build_pair
tmp = pathlib.Path(tempfile.mkdtemp())
X, o, h = build_pair(native, oracle, tmp, 6000, 128, 16, 100, "DistL2", seed=6144)
Q = uniform(701, 128, 5)
ref = o.parallel_search(Q, 10, 48)
ids_p, d_p, cnt_p, st_p = _device_call_with_stats(native, h, Q, 10, 48)
os.environ["HNSWGPU_NO_PAIR_DESCENT"] = "1"; native.reload_env()
ids_s, d_s, cnt_s, st_s = _device_call_with_stats(native, h, Q, 10, 48)
bad = np.where((ids_p != ref.ids.astype(np.uint64)).any(axis=1))[0]
print("mismatching queries", len(bad), bad[:40])
print("single ok", np.array_equal(ids_s, ref.ids.astype(np.uint64)))
diff7 = np.where(st_p[:, 7] != st_s[:, 7])[0]
print("word7 differs", len(diff7), diff7[:40])
for q in diff7[:10]:
    print(q, "pair n_exp,n_dist", (st_p[q, 7] >> 8) & 0xFF, st_p[q, 7] >> 16, "single", (st_s[q, 7] >> 8) & 0xFF, st_s[q, 7] >> 16)
print("entry level", h.get_max_level_observed())
