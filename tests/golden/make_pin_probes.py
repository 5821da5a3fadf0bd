#!/usr/bin/env python3
"""Generates the PROBES that let the pin localise a mismatch (oracle/PIN.md): the day oracle/ref_pin runs with the real crate,
a failing pin should say "the arithmetic of DistCosine at d = 25" or "the pop order of BinaryHeap", not just "ids differ".
  pin_pairs.bin        : fixed vector pairs: for each (metric, d) block a header {u32 metric, u32 d, u32 n} and n x 2 x d f32
                         (metric ids of include/hnsw_mi355x.h: 0 L2, 1 Cosine, 2 Dot, 3 L1, 4 Hellinger, 5 Jeffreys, 6 JensenShannon;
                         d in 1, 3, 25, 128, 784; Dot on l2-normalised vectors, the probability distances on probability vectors)
  pin_pairs.npz        : the oracle's eval of every pair (f32 bit patterns), scalar order -- what `D::default().eval(a, b)` must give
  pin_heap_scripts.bin : tie-heavy scripts for std::collections::BinaryHeap with a distance-only Ord (src/hnsw.rs:283-297):
                         per script {u32 n_ops} then n_ops x {u8 is_pop, f32 value, i32 tag}; after the ops: into_sorted_vec
  pin_heap_scripts.npz : the oracle's popped tags per script and the tags of into_sorted_vec of what is left
Run from the repo root:  python tests/golden/make_pin_probes.py"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402

METRICS = ["DistL2", "DistCosine", "DistDot", "DistL1", "DistHellinger", "DistJeffreys", "DistJensenShannon"]
DIMS = [1, 3, 25, 128, 784]
N_PAIRS = {1: 200, 3: 200, 25: 200, 128: 60, 784: 16}  # pairs per dimension (the file stays ~1.4 MB)


def pairs_for(metric, d, rng):
    n = N_PAIRS[d]
    a = rng.random((n, d), dtype=np.float32)
    b = rng.random((n, d), dtype=np.float32)
    if metric in ("DistL2", "DistL1"):
        a -= np.float32(0.5)
        b[::7] = a[::7]                      # identical vectors: distance exactly 0
    if metric == "DistCosine":
        a[::9] *= np.float32(1e10)           # norms far apart
        b[5] = 0.0                           # the zero-norm rule
    if metric == "DistDot":
        for m in (a, b):
            for i in range(n):
                oracle_lib.lib().orc_l2_normalize(m[i].ctypes.data, d)
        b[::7] = a[::7]                      # 1 - 1 (+- rounding): the clamp at 0
    if metric in ("DistHellinger", "DistJeffreys", "DistJensenShannon"):
        for m in (a, b):
            m += np.float32(1e-3)
            if d > 2:
                m[:, ::5] = 0.0              # exact zeros (the M_MIN / skip rules)
            m /= m.sum(1, dtype=np.float32)[:, None]
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def main():
    rng = np.random.default_rng(0x9117)
    blob = bytearray()
    expect = {}
    for mi, metric in enumerate(METRICS):
        for d in DIMS:
            a, b = pairs_for(metric, d, rng)
            blob += struct.pack("<III", mi, d, len(a))
            blob += np.stack([a, b], axis=1).astype("<f4").tobytes()   # pair i: a_i then b_i
            out = np.array([oracle_lib.dist_matrix(metric, a[i:i + 1], b[i:i + 1])[0, 0] for i in range(len(a))], np.float32)
            expect[f"{metric}_d{d}"] = out.view(np.uint32)
    open(os.path.join(HERE, "pin_pairs.bin"), "wb").write(bytes(blob))
    np.savez_compressed(os.path.join(HERE, "pin_pairs.npz"), **expect)
    # heap scripts: few distinct values (ties everywhere), interleaved pushes and pops, lengths around the sizes the search uses
    blob = bytearray()
    popped, left = {}, {}
    for si, (n_ops, n_vals, p_pop) in enumerate([(60, 3, 0.3), (200, 5, 0.4), (400, 8, 0.45), (130, 2, 0.2), (700, 16, 0.35), (65, 1, 0.5)]):
        vals = rng.integers(0, n_vals, n_ops).astype(np.float32)
        is_pop = (rng.random(n_ops) < p_pop).astype(np.uint8)
        is_pop[:4] = 0
        tags = np.arange(n_ops, dtype=np.int32)
        blob += struct.pack("<I", n_ops)
        for i in range(n_ops):
            blob += struct.pack("<Bfi", int(is_pop[i]), float(vals[i]), int(tags[i]))
        _pv, pt, _sv, st = oracle_lib.heap_script(vals, tags, is_pop)
        popped[f"s{si}"] = pt.astype(np.int32)
        left[f"s{si}"] = st.astype(np.int32)
    open(os.path.join(HERE, "pin_heap_scripts.bin"), "wb").write(bytes(blob))
    np.savez_compressed(os.path.join(HERE, "pin_heap_scripts.npz"), **{"popped_" + k: v for k, v in popped.items()},
                        **{"sorted_" + k: v for k, v in left.items()})
    print("wrote pin_pairs.{bin,npz}, pin_heap_scripts.{bin,npz}")


if __name__ == "__main__":
    main()
