#!/usr/bin/env python3
"""Generates the golden fixtures of this directory with the CPU oracle.

The reference (Rust) cannot be built or imported in this environment and its tests hold no known-answer
vectors, so these fixtures pin the ORACLE's outputs (and, through it, the product's) across changes:
  <name>.hnsw.graph / .hnsw.data : an index built by the oracle's restated serial insert, in hnswio format
  <name>.npz                     : seeded queries + the oracle's answers (ids, f32 distance bits, p_ids, counts), plus a
                                   sorted-id filter and the oracle's search_filter answers (f_* arrays; count
                                   0xFFFFFFFF = the reference panics on that query, src/hnsw.rs:973)
  <name>.queries.bin / .filter.bin : the same queries / filter as raw little-endian files for oracle/ref_pin (the Rust
                                   program that answers them with the REAL reference: see oracle/PIN.md)
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402

CASES = {
    # name: n, d, M, ef_c, dist, normalize, k, ef, nq
    "l2_d25": (240, 25, 8, 40, "DistL2", False, 10, 24, 40),
    "l2_d128": (160, 128, 8, 40, "DistL2", False, 10, 64, 24),
    "cos_d25": (200, 25, 8, 40, "DistCosine", False, 10, 32, 32),
    "dot_d25": (200, 25, 8, 40, "DistDot", True, 5, 16, 32),
    "l1_d10": (200, 10, 10, 25, "DistL1", False, 10, 20, 32),
    "hell_d12": (180, 12, 8, 30, "DistHellinger", "prob", 8, 24, 24),
    "jeff_d12": (180, 12, 8, 30, "DistJeffreys", "prob", 8, 24, 24),
    "js_d12": (180, 12, 8, 30, "DistJensenShannon", "prob", 8, 24, 24),
    # tie-saturated data: what these pin, the day oracle/ref_pin runs, is the ORDER of std's BinaryHeap among equal
    # distances (pop order, "pop when len > ef", into_sorted_vec), not only tie-free arithmetic
    "l2_dup_d16": (300, 16, 8, 40, "DistL2", "dup", 10, 32, 40),     # every vector stored three times under distinct ids
    "l1_grid_d4": (256, 4, 6, 30, "DistL1", "grid", 10, 24, 40),     # small-integer grid (tests/filtertest.rs:229-241): integer L1 distances
    "l2_ef100_d32": (400, 32, 12, 60, "DistL2", False, 100, 100, 16),  # ef > 64 (two result slots per lane), k = ef: all of return_points, sorted
}


def main():
    for name, (n, d, m, efc, dist, normalize, k, ef, nq) in CASES.items():
        rng = np.random.default_rng(abs(hash(name)) % (1 << 31) if False else sum(map(ord, name)))
        X = rng.random((n, d), dtype=np.float32)
        Q = rng.random((nq, d), dtype=np.float32)
        if normalize == "dup":      # n / 3 distinct vectors, each three times; half of the queries ARE stored vectors
            X = np.ascontiguousarray(np.tile(X[: n // 3], (3, 1)))[rng.permutation(n)]
            Q[::2] = X[rng.choice(n, len(Q[::2]), replace=False)]
        elif normalize == "grid":   # coordinates in {0, 1, 2, 3}: every distance is a small integer, ties everywhere
            X = rng.integers(0, 4, (n, d)).astype(np.float32)
            Q = rng.integers(0, 4, (nq, d)).astype(np.float32)
        if normalize == "prob":   # probability vectors (with exact zeros)
            for a in (X, Q):
                a += np.float32(1e-3)
                a[:, ::5] = 0.0
                a /= a.sum(1, dtype=np.float32)[:, None]
        elif normalize is True:
            for a in (X, Q):
                for i in range(a.shape[0]):
                    oracle_lib.lib().orc_l2_normalize(a[i].ctypes.data, d)
        o = oracle_lib.OracleHnsw(m, n, 16, efc, dist)
        o.insert_batch(X, ids=np.arange(n) * 3 + 1)
        o.file_dump(HERE, name)
        r = o.parallel_search(Q, k, ef, 1)
        # filtered search (impl FilterT for Vec<usize>): every third origin id of a contiguous stretch, plus a few ids
        # that are not in the index
        origin = np.arange(n, dtype=np.uint64) * 3 + 1
        allowed = np.sort(np.concatenate([origin[n // 4:n // 2:3], np.array([0, 2, 5, 3 * n + 7], np.uint64)]))
        f_ids = np.zeros((nq, k), np.uint64)
        f_bits = np.zeros((nq, k), np.uint32)
        f_layers = np.zeros((nq, k), np.uint8)
        f_ranks = np.zeros((nq, k), np.int32)
        f_counts = np.zeros(nq, np.uint32)
        for i in range(nq):
            try:
                ids, dd, ll, rr = o.search_filter(Q[i], k, ef, allowed)
            except RuntimeError as e:
                assert "panics" in str(e), e
                f_counts[i] = 0xFFFFFFFF
                continue
            c = len(ids)
            f_counts[i] = c
            f_ids[i, :c], f_bits[i, :c], f_layers[i, :c], f_ranks[i, :c] = ids, dd.view(np.uint32), ll, rr
        np.savez_compressed(os.path.join(HERE, name + ".npz"), queries=Q, k=k, ef=ef, dist=dist, ids=r.ids,
                            dist_bits=r.dists.view(np.uint32), layers=r.layers, ranks=r.ranks, counts=r.counts,
                            filter_ids=allowed, f_ids=f_ids, f_dist_bits=f_bits, f_layers=f_layers, f_ranks=f_ranks,
                            f_counts=f_counts)
        with open(os.path.join(HERE, name + ".queries.bin"), "wb") as f:
            f.write(np.array([nq, d, k, ef], "<u4").tobytes())
            f.write(Q.astype("<f4").tobytes())
        with open(os.path.join(HERE, name + ".filter.bin"), "wb") as f:
            f.write(allowed.astype("<u8").tobytes())
        print(name, "points", n, "queries", nq, "graph bytes", os.path.getsize(os.path.join(HERE, name + ".hnsw.graph")))


if __name__ == "__main__":
    main()
