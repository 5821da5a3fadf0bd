"""A second, independent transcription of the reference's search -- `Hnsw::search_filter` with no filter (src/hnsw.rs:1487-1580)
and `search_layer` (src/hnsw.rs:922-1064) -- in plain Python, kept apart from oracle/hnsw_oracle.hpp on purpose: it walks the
graph of a committed dump through the product's host-side accessors (no device), keeps both `BinaryHeap`s with the Python
transcription of std's heap from tests/test_oracle.py, sums distances in numpy float32 left to right (`np.add.accumulate`), and must
reproduce the committed answers (ids, f32 distance bits, p_ids, counts) -- on tie-free data and on the tie-saturated fixtures,
where the answer depends on the heaps' internal order.  Not a pin (the fixtures were written by the oracle), but two
transcriptions of the same Rust text, by different routes, have to agree."""
import os

import numpy as np
import pytest

from test_oracle import _StdBinaryHeap

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32


def _l2(a, b):
    t = a.astype(f32) - b.astype(f32)
    # (np.add.accumulate runs left to right in float32: the reference's scalar sum, one rounding per element; np.sum would pair up)
    return f32(np.sqrt(np.add.accumulate(t * t, dtype=f32)[-1]))


def _l1(a, b):
    return np.add.accumulate(np.abs(a.astype(f32) - b.astype(f32)), dtype=f32)[-1]


class _Graph:
    """points by PointId = (layer, rank); lists in stored order; vectors by origin id"""

    def __init__(self, native, name, dist, directory=GOLD):
        self.h = native.HnswIo(directory, name).load_hnsw(dist)
        self.dm = native.DataMap.from_hnswdump(directory, name)
        self.eval = _l2 if dist == "DistL2" else _l1
        self.origin = {}
        o, pid = self.h.get_entry_point()
        self.entry = pid
        self.origin[pid] = o
        self._lists = {}

    def neighbours(self, pid, layer):
        key = (pid, layer)
        if key not in self._lists:
            ids, layers, ranks, _ = self.h.get_neighbours(pid[0], pid[1], layer)
            lst = [(int(l), int(r)) for l, r in zip(layers, ranks)]
            for p, o in zip(lst, ids):
                self.origin[p] = int(o)
            self._lists[key] = lst
        return self._lists[key]

    def dist(self, q, pid):
        return self.eval(q, self.dm.get_data(self.origin[pid]))


def _search_layer(g, q, entry, ef, layer):
    """src/hnsw.rs:922-1064, filter = None.  Heaps hold (key, PointId); BinaryHeap compares keys only."""
    ret, cand = _StdBinaryHeap(), _StdBinaryHeap()
    if g.h.get_layer_nb_point(layer) == 0:           # :942-946
        return ret
    d0 = g.dist(q, entry)                             # :952
    visited = {entry}                                 # :955-956
    cand.push((-float(d0), entry))                    # :958-963
    ret.push((float(d0), entry))                      # :964-967
    while cand.d:                                     # :969
        c = cand.pop()                                # :971
        f = ret.d[0]                                  # peek: the root
        if -c[0] > f[0]:                              # :981
            return ret                                # :993
        for e in g.neighbours(c[1], layer):           # :1005-1014
            if e in visited:                          # :1015
                continue
            visited.add(e)                            # :1016
            if not ret.d:
                return ret
            fd = ret.d[0][0]
            de = float(g.dist(q, e))                  # :1026
            if de < fd or len(ret.d) < ef:            # :1028
                cand.push((-de, e))                   # :1035-1036
                ret.push((de, e))                     # :1038
                if len(ret.d) > ef:                   # :1051-1053
                    ret.pop()
    return ret


def _search(g, q, knbn, ef_arg):
    """src/hnsw.rs:1487-1580, filter = None"""
    entry = g.entry
    dist_to_entry = g.dist(q, entry)
    pivot = entry
    for layer in range(entry[0], 0, -1):              # :1510: (1..=entry_point.p_id.0).rev()
        new_pivot, changed = None, False
        for n in g.neighbours(pivot, layer):
            tmp = g.dist(q, n)
            if tmp < dist_to_entry:
                new_pivot, changed, dist_to_entry = n, True, tmp
        if changed:
            pivot = new_pivot
    ef = max(ef_arg, knbn)                            # :1531
    layer = 0
    while g.h.get_layer_nb_point(layer) == 0:         # :1534-1540
        layer += 1
    heap = _search_layer(g, q, pivot, ef, layer)
    srt = heap.into_sorted_vec()                      # :1544
    last = min(knbn, ef, len(srt))                    # :1547
    return [(g.origin[p], f32(d), p) for d, p in srt[:last]]


@pytest.mark.parametrize("name", ["l2_d25", "l2_dup_d16", "l1_grid_d4", "l2_ef100_d32"])
def test_python_transcription_reproduces_the_committed_answers(native, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    k, ef, dist = int(z["k"]), int(z["ef"]), str(z["dist"])
    g = _Graph(native, name, dist)
    nq = min(len(z["queries"]), 16 if name == "l2_ef100_d32" else 40)
    ties = 0
    for i in range(nq):
        got = _search(g, z["queries"][i], k, ef)
        cnt = int(z["counts"][i])
        assert len(got) == cnt, (name, i)
        ids = [o for o, _, _ in got]
        bits = np.array([d for _, d, _ in got], f32).view(np.uint32)
        assert ids == [int(v) for v in z["ids"][i][:cnt]], (name, i)
        assert np.array_equal(bits, z["dist_bits"][i][:cnt]), (name, i)
        assert [p[0] for _, _, p in got] == [int(v) for v in z["layers"][i][:cnt]]
        assert [p[1] for _, _, p in got] == [int(v) for v in z["ranks"][i][:cnt]]
        ties += int(len(set(bits.tolist())) < cnt)
    if name in ("l2_dup_d16", "l1_grid_d4"):
        assert ties > nq // 2   # these fixtures really are decided by heap order: equal distances inside the answers


class _HeapWithRetain(_StdBinaryHeap):
    """+ `BinaryHeap::retain` (std >= 1.70): `Vec::retain` keeps the survivors in order and notes the index of the first entry
    removed; then `rebuild_tail(first_removed)`: nothing if no entry went, a full `rebuild` (sift_down of every internal node,
    last first) when the tail is the larger part or cheaper by std's estimate, else `sift_up` of every tail entry in order."""

    def _sift_down_range(self, pos, end):
        d = self.d
        elt = d[pos]
        child = 2 * pos + 1
        while end >= 2 and child <= end - 2:
            if d[child][0] <= d[child + 1][0]:
                child += 1
            if elt[0] >= d[child][0]:
                d[pos] = elt
                return
            d[pos] = d[child]
            pos = child
            child = 2 * pos + 1
        if child == end - 1 and elt[0] < d[child][0]:
            d[pos] = d[child]
            pos = child
        d[pos] = elt

    def retain(self, keep):
        first_removed = len(self.d)
        out = []
        for i, e in enumerate(self.d):
            if keep(e):
                out.append(e)
            elif i < first_removed:
                first_removed = i
        self.d = out
        n, start = len(out), first_removed
        if start >= n:
            return
        tail = n - start
        log2 = start.bit_length() - 1 if start > 0 else 0
        if start < tail:
            rebuild = True
        elif n <= 2048:
            rebuild = 2 * n < tail * log2
        else:
            rebuild = 2 * n < tail * 11
        if rebuild:
            for node in range(n // 2 - 1, -1, -1):
                self._sift_down_range(node, n)
        else:
            for i in range(start, n):
                self._sift_up(0, i)


class _ReferencePanic(Exception):
    pass


def _search_layer_filtered(g, q, entry, ef, layer, allowed):
    """src/hnsw.rs:922-1064 with a filter (`impl FilterT for Vec<usize>`: binary search of the origin id, src/filter.rs:11-15)"""
    ok = lambda pid: g.origin[pid] in allowed  # noqa: E731
    ret, cand = _HeapWithRetain(), _HeapWithRetain()
    if g.h.get_layer_nb_point(layer) == 0:
        return ret
    d0 = float(g.dist(q, entry))
    visited = {entry}
    cand.push((-d0, entry))
    ret.push((d0, entry))                              # (the entry point goes in whether the filter allows it or not, :964-967)
    while cand.d:
        c = cand.pop()
        if not ret.d:
            raise _ReferencePanic()                    # :973 `return_points.peek().unwrap()` on an emptied heap
        f = ret.d[0]
        if -c[0] > f[0] and len(ret.d) >= ef:          # :981, :994-1000: no return with a filter
            ret.retain(lambda e: ok(e[1]))
        for e in g.neighbours(c[1], layer):
            if e in visited:
                continue
            visited.add(e)
            if not ret.d:
                return ret                             # :1019-1024
            fd = ret.d[0][0]
            de = float(g.dist(q, e))
            if de < fd or len(ret.d) < ef:
                cand.push((-de, e))
                if ok(e):                              # :1040-1049
                    if len(ret.d) == 1 and not ok(ret.d[0][1]):
                        ret.d = []
                    ret.push((de, e))
                if len(ret.d) > ef:
                    ret.pop()
    return ret


def _search_filtered(g, q, knbn, ef_arg, allowed):
    entry = g.entry
    dist_to_entry = g.dist(q, entry)
    pivot = entry
    for layer in range(entry[0], 0, -1):
        new_pivot, changed = None, False
        for n in g.neighbours(pivot, layer):
            tmp = g.dist(q, n)
            if tmp < dist_to_entry:
                new_pivot, changed, dist_to_entry = n, True, tmp
        if changed:
            pivot = new_pivot
    ef = max(ef_arg, knbn)
    layer = 0
    while g.h.get_layer_nb_point(layer) == 0:
        layer += 1
    srt = _search_layer_filtered(g, q, pivot, ef, layer, allowed).into_sorted_vec()
    last = min(knbn, ef, len(srt))
    return [(g.origin[p], f32(d), p) for d, p in srt[:last] if g.origin[p] in allowed]   # :1549-1565


@pytest.mark.parametrize("name", ["l2_d25", "l2_dup_d16", "l1_grid_d4"])
def test_python_transcription_reproduces_the_committed_filtered_answers(native, name):
    """`Hnsw::search_filter` with a sorted id vector: the loop that never returns early, `retain`, the `clear` of a lone refused
    entry, the final `filter_map` -- and the panic of the reference on an emptied `return_points` (count 0xFFFFFFFF)."""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    k, ef, dist = int(z["k"]), int(z["ef"]), str(z["dist"])
    g = _Graph(native, name, dist)
    allowed = set(int(v) for v in z["filter_ids"])
    for i in range(len(z["queries"])):
        want = int(z["f_counts"][i])
        try:
            got = _search_filtered(g, z["queries"][i], k, ef, allowed)
        except _ReferencePanic:
            assert want == 0xFFFFFFFF, (name, i)
            continue
        assert want != 0xFFFFFFFF and len(got) == want, (name, i, want, len(got))
        assert [o for o, _, _ in got] == [int(v) for v in z["f_ids"][i][:want]], (name, i)
        assert np.array_equal(np.array([d for _, d, _ in got], f32).view(np.uint32), z["f_dist_bits"][i][:want]), (name, i)
        assert [p[1] for _, _, p in got] == [int(v) for v in z["f_ranks"][i][:want]]


@pytest.mark.parametrize("density,k,ef", [(0.004, 8, 12), (0.02, 8, 12), (0.3, 8, 12), (0.9, 8, 12), (0.05, 1, 1), (0.3, 1, 1), (0.5, 2, 2), (0.0, 1, 1), (0.0, 2, 2), (0.0, 8, 12)])
def test_python_transcription_against_the_oracle_under_filters_of_all_densities(native, oracle, tmp_path, density, k, ef):
    """Sparse filters make the reference panic (an emptied `return_points`), dense ones reach `retain` with the refused entry
    point still inside: oracle and Python transcription must agree query by query, panics included."""
    from conftest import uniform
    n, d = 400, 6
    X = uniform(n, d, 91)
    X[n // 2:] = X[: n // 2]                       # every point twice: ties on top
    o = oracle.OracleHnsw(6, n, 16, 30, "DistL2")
    o.insert_batch(X)
    o.file_dump(tmp_path, "flt")
    g = _Graph(native, "flt", "DistL2", tmp_path)
    rng = np.random.default_rng(int(density * 1000) + 7 * ef)
    allowed = np.sort(rng.choice(n, int(n * density), replace=False)).astype(np.uint64)   # (density 0: nothing is allowed)
    aset = set(int(v) for v in allowed)
    Q = uniform(60, d, 92)
    panics = answered = 0
    for i in range(len(Q)):
        try:
            ref = o.search_filter(Q[i], k, ef, allowed)      # (ids, dists, layers, ranks)
        except RuntimeError as e:                            # the oracle reports the reference's panic as an error
            assert "panics" in str(e), e
            ref = None
        try:
            got = _search_filtered(g, Q[i], k, ef, aset)
        except _ReferencePanic:
            assert ref is None, i
            panics += 1
            continue
        assert ref is not None, i
        answered += 1
        assert [o_ for o_, _, _ in got] == [int(v) for v in ref[0]], i
        assert np.array_equal(np.array([d_ for _, d_, _ in got], f32).view(np.uint32), ref[1].view(np.uint32)), i
        assert [p[1] for _, _, p in got] == [int(v) for v in ref[3]], i
    assert answered + panics == len(Q)
