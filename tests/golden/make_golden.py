#!/usr/bin/env python3
"""Generates the golden fixtures of this directory with the CPU oracle.

The reference (Rust) cannot be built or imported in this environment and its tests hold no known-answer
vectors, so these fixtures pin the ORACLE's outputs (and, through it, the product's) across changes:
  <name>.hnsw.graph / .hnsw.data : an index built by the oracle's restated serial insert, in hnswio format
  <name>.npz                     : seeded queries + the oracle's answers (ids, f32 distance bits, p_ids, counts)
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
}


def main():
    for name, (n, d, m, efc, dist, normalize, k, ef, nq) in CASES.items():
        rng = np.random.default_rng(abs(hash(name)) % (1 << 31) if False else sum(map(ord, name)))
        X = rng.random((n, d), dtype=np.float32)
        Q = rng.random((nq, d), dtype=np.float32)
        if normalize:
            for a in (X, Q):
                for i in range(a.shape[0]):
                    oracle_lib.lib().orc_l2_normalize(a[i].ctypes.data, d)
        o = oracle_lib.OracleHnsw(m, n, 16, efc, dist)
        o.insert_batch(X, ids=np.arange(n) * 3 + 1)
        o.file_dump(HERE, name)
        r = o.parallel_search(Q, k, ef, 1)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), queries=Q, k=k, ef=ef, dist=dist, ids=r.ids,
                            dist_bits=r.dists.view(np.uint32), layers=r.layers, ranks=r.ranks, counts=r.counts)
        print(name, "points", n, "queries", nq, "graph bytes", os.path.getsize(os.path.join(HERE, name + ".hnsw.graph")))


if __name__ == "__main__":
    main()
