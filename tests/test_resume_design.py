"""Design check (CPU) for the HNSW_STRICT_RESUME experiment of the search kernel (search_kernels.inc): the first attempt
keeps return_points / candidate_points as ONE sorted array and logs its heap operations; at the first insertion that
meets an equal distance the two literal BinaryHeaps are rebuilt from the log and the search carries on with them --
the rest of the interrupted batch first.  The claim: the answer equals the reference's (the oracle's) literal search
also on data where nearly every query meets ties.  This file restates that control flow in Python (same batch
structure, same hand-over point) on graphs read back through the product's host API, and compares with the oracle."""
import numpy as np
import pytest

EXPAND = None


# ---- literal std::collections::BinaryHeap on (key, payload) pairs, compared by key only (checked against the oracle below)
def _sift_up(d, pos):
    e = d[pos]
    while pos > 0:
        par = (pos - 1) // 2
        if e[0] <= d[par][0]:
            break
        d[pos] = d[par]
        pos = par
    d[pos] = e


def heap_push(d, x):
    d.append(x)
    _sift_up(d, len(d) - 1)


def heap_pop(d):
    item = d.pop()
    if d:
        item, d[0] = d[0], item
        end, pos, e, child = len(d), 0, d[0], 1
        while end >= 2 and child <= end - 2:
            if d[child][0] <= d[child + 1][0]:
                child += 1
            d[pos] = d[child]
            pos = child
            child = 2 * pos + 1
        if child == end - 1:
            d[pos] = d[child]
            pos = child
        d[pos] = e
        _sift_up(d, pos)
    return item


def heap_into_sorted(d):
    d = list(d)
    end = len(d)
    while end > 1:
        end -= 1
        d[0], d[end] = d[end], d[0]
        pos, e, child = 0, d[0], 1
        while end >= 2 and child <= end - 2:
            if d[child][0] <= d[child + 1][0]:
                child += 1
            if e[0] >= d[child][0]:
                break
            d[pos] = d[child]
            pos = child
            child = 2 * pos + 1
        else:
            if child == end - 1 and e[0] < d[child][0]:
                d[pos] = d[child]
                pos = child
        d[pos] = e
    return d


def test_python_heap_matches_the_oracle(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        n = int(rng.integers(5, 400))
        vals = rng.integers(0, 5, n).astype(np.float32)
        pops = (rng.random(n) < 0.35).astype(np.uint8)
        pv, pt, sv, st = oracle.heap_script(vals, np.arange(n), pops)
        d, got = [], []
        for i in range(n):
            if pops[i]:
                if d:
                    got.append(heap_pop(d)[1])
            else:
                heap_push(d, (float(vals[i]), i))
        assert got == pt.tolist()
        assert [e[1] for e in heap_into_sorted(d)] == st.tolist()


class Graph:
    """Layer-0 lists and vectors of an index, through the product's host getters."""

    def __init__(self, h, X, dist, oracle):
        self.h, self.X, self.dist, self.oracle = h, X, dist, oracle
        self.cache = {}

    def nbrs(self, p, layer):          # p = (origin, layer, rank)
        key = (p, layer)
        if key not in self.cache:
            ids, layers, ranks, _ = self.h.get_neighbours(p[1], p[2], layer)
            self.cache[key] = [(int(i), int(l), int(r)) for i, l, r in zip(ids, layers, ranks)]
        return self.cache[key]

    def d(self, q, p):
        return self.oracle.dist_eval(self.dist, q, self.X[p[0]])


def descent(g, q):
    origin, (layer, rank) = g.h.get_entry_point()
    pivot = (origin, layer, rank)
    dcur = g.d(q, pivot)
    for l in range(layer, 0, -1):
        best, best_p = dcur, None
        for nb in g.nbrs(pivot, l):
            dd = g.d(q, nb)
            if dd < best:
                best, best_p = dd, nb
        if best_p is not None:
            dcur, pivot = best, best_p
    return pivot, dcur


def search_with_resume(g, q, k, ef, batch=64):
    """Returns (answer ids, resumed?)."""
    pivot, dcur = descent(g, q)
    R = [[dcur, pivot, False]]          # sorted ascending, arrival order among equals; flag = expanded
    visited = {pivot}
    log = []
    state = None                        # set at the hand-over: (candidate heap, result heap)

    def accept_exact(C, RR, cands):
        for xd, p in cands:
            if xd < RR[0][0] or len(RR) < ef:
                heap_push(C, (-xd, p))
                heap_push(RR, (xd, p))
                if len(RR) > ef:
                    heap_pop(RR)

    while state is None:
        cj = next((j for j, e in enumerate(R) if not e[2]), None)
        if cj is None:
            break
        R[cj][2] = True
        c = R[cj][1]
        log.append(EXPAND)
        lst = g.nbrs(c, 0)
        for b0 in range(0, len(lst), batch):
            fresh = [p for p in lst[b0:b0 + batch] if p not in visited]
            visited.update(fresh)
            de = [(g.d(q, p), p) for p in fresh]
            worst = R[-1][0]
            cand = [x for x in de if len(R) < ef or x[0] < worst]     # the ballot at the start of the batch
            for i, (xd, p) in enumerate(cand):
                if xd < worst or len(R) < ef:
                    log.append((xd, p))
                    tie = any(e[0] == xd for e in R)
                    pos = sum(1 for e in R if e[0] <= xd)
                    R.insert(pos, [xd, p, False])
                    if len(R) > ef:
                        R.pop()
                    if tie:
                        # hand over: rebuild both heaps from the log ...
                        C, RR = [], []
                        heap_push(C, (-dcur, pivot))
                        heap_push(RR, (dcur, pivot))
                        for op in log:
                            if op is EXPAND:
                                heap_pop(C)
                            else:
                                heap_push(C, (-op[0], op[1]))
                                heap_push(RR, op)
                                if len(RR) > ef:
                                    heap_pop(RR)
                        # ... offer the rest of this batch, then the remaining batches of this expansion
                        accept_exact(C, RR, cand[i + 1:])
                        for b1 in range(b0 + batch, len(lst), batch):
                            fresh = [p2 for p2 in lst[b1:b1 + batch] if p2 not in visited]
                            visited.update(fresh)
                            de2 = [(g.d(q, p2), p2) for p2 in fresh]
                            w = RR[0][0]
                            accept_exact(C, RR, [x for x in de2 if len(RR) < ef or x[0] < w])
                        state = (C, RR)
                        break
                    worst = R[-1][0]
            if state is not None:
                break
    if state is None:
        return [e[1][0] for e in R[:k]], False
    C, RR = state
    while C:
        ce = heap_pop(C)
        if -ce[0] > RR[0][0]:
            break
        for b0 in range(0, len(g.nbrs(ce[1], 0)), batch):
            fresh = [p for p in g.nbrs(ce[1], 0)[b0:b0 + batch] if p not in visited]
            visited.update(fresh)
            de = [(g.d(q, p), p) for p in fresh]
            w = RR[0][0]
            accept_exact(C, RR, [x for x in de if len(RR) < ef or x[0] < w])
    out = heap_into_sorted(RR)
    return [e[1][0] for e in out[:min(k, ef)]], True


@pytest.mark.parametrize("kind,dist,ef,m", [("grid", "DistL2", 16, 8), ("duplicates", "DistL2", 24, 8), ("grid", "DistL1", 40, 40),
                                             ("uniform", "DistL2", 32, 12)])
def test_resume_from_log_equals_the_literal_search(native, oracle, tmp_path, kind, dist, ef, m):
    rng = np.random.default_rng(91)
    n, d = 1200, 6
    if kind == "duplicates":
        base = rng.random((n // 2, d), dtype=np.float32)
        X = np.concatenate([base, base])[rng.permutation(n)]
    elif kind == "grid":
        X = rng.integers(0, 4, (n, d)).astype(np.float32)
    else:
        X = rng.random((n, d), dtype=np.float32)
    X = np.ascontiguousarray(X)
    o = oracle.OracleHnsw(m, n, 16, 60, dist)
    o.insert_batch(X)
    o.file_dump(tmp_path, "r")
    h = native.HnswIo(tmp_path, "r").load_hnsw(dist)
    g = Graph(h, X, dist, oracle)
    resumed = 0
    for qi in range(40):
        q = rng.integers(0, 4, d).astype(np.float32) if kind == "grid" else rng.random(d, dtype=np.float32)
        ids, was = search_with_resume(g, q, 10, ef, batch=16)   # M=40 lists span several batches of 16
        resumed += was
        ref_ids, _, _, _ = o.search(q, 10, ef)
        assert ids == ref_ids.tolist(), f"query {qi} ({'resumed' if was else 'tie free'})"
    if kind != "uniform":
        assert resumed > 10    # the tie-heavy data sets do exercise the hand-over


def search_literal_c_value_r(g, q, k, ef):
    """Second design check.  Only candidate_points needs its literal heap during the search: return_points influences
    the trajectory through its largest VALUE and its length alone, so it can stay a value-sorted array, and its
    literal heap is rebuilt from the log of its operations at the END -- and only if a choice between equal entries
    can have reached the answer: an eviction between two equal farthest entries whose survivor is still in the top k,
    or equal neighbours inside the answer (or across its end).  Returns (ids, replayed?)."""
    pivot, dcur = descent(g, q)
    C = []
    heap_push(C, (-dcur, pivot))
    R = [[dcur, pivot, False]]          # [distance, point, tainted]
    rlog = [(dcur, pivot)]
    visited = {pivot}
    while C:
        ce = heap_pop(C)
        if -ce[0] > R[-1][0]:
            break
        for p in g.nbrs(ce[1], 0):
            if p in visited:
                continue
            visited.add(p)
            xd = g.d(q, p)
            if xd < R[-1][0] or len(R) < ef:
                heap_push(C, (-xd, p))
                rlog.append((xd, p))
                pos = sum(1 for e in R if e[0] <= xd)
                R.insert(pos, [xd, p, False])
                if len(R) > ef:
                    if R[-1][0] == R[-2][0]:     # which of the equal farthest entries leaves depends on the heap layout
                        R[-2][2] = True          # ... so whoever stays is tainted (all entries at that distance)
                        for e in R:
                            if e[0] == R[-1][0]:
                                e[2] = True
                    R.pop()
    kk = min(k, ef, len(R))
    need = any(e[2] for e in R[:kk]) or any(R[j][0] == R[j + 1][0] for j in range(kk) if j + 1 < len(R))
    if not need:
        return [e[1][0] for e in R[:kk]], False
    RR = []
    for op in rlog:
        heap_push(RR, op)
        if len(RR) > ef:
            heap_pop(RR)
    return [e[1][0] for e in heap_into_sorted(RR)[:kk]], True


@pytest.mark.parametrize("kind,dist,ef,m", [("grid", "DistL2", 16, 8), ("duplicates", "DistL2", 24, 8), ("grid", "DistL1", 40, 40),
                                             ("uniform", "DistL2", 32, 12), ("grid", "DistL2", 12, 6)])
def test_value_sorted_result_set_with_literal_candidate_heap(native, oracle, tmp_path, kind, dist, ef, m):
    rng = np.random.default_rng(17)
    n, d = 1200, 6
    if kind == "duplicates":
        base = rng.random((n // 2, d), dtype=np.float32)
        X = np.concatenate([base, base])[rng.permutation(n)]
    elif kind == "grid":
        X = rng.integers(0, 4, (n, d)).astype(np.float32)
    else:
        X = rng.random((n, d), dtype=np.float32)
    X = np.ascontiguousarray(X)
    o = oracle.OracleHnsw(m, n, 16, 60, dist)
    o.insert_batch(X)
    o.file_dump(tmp_path, "v")
    h = native.HnswIo(tmp_path, "v").load_hnsw(dist)
    g = Graph(h, X, dist, oracle)
    replayed = 0
    for qi in range(60):
        q = rng.integers(0, 4, d).astype(np.float32) if kind == "grid" else rng.random(d, dtype=np.float32)
        ids, was = search_literal_c_value_r(g, q, 10, ef)
        replayed += was
        ref_ids, _, _, _ = o.search(q, 10, ef)
        assert ids == ref_ids.tolist(), f"query {qi} ({'R replayed' if was else 'R by value'})"
    print(f"{kind} {dist} ef={ef}: return_points replayed for {replayed} of 60 queries")


def search_end_state(g, q, k, ef, batch=16):
    """Both experiments together (the intended end state): first attempt with a log; at the first equal pair only
    candidate_points is rebuilt as a literal heap, return_points simply stays the value-sorted array it already is
    (taint tracking starts there: before the first equal pair no eviction can have been ambiguous), and the log keeps
    growing so that return_points can be rebuilt literally at the end if an equal-entry choice reached the answer."""
    pivot, dcur = descent(g, q)
    R = [[dcur, pivot, False, False]]     # distance, point, expanded (first attempt only), tainted
    visited = {pivot}
    log = []
    C = None

    def offer(cands):                     # literal candidate heap + value-sorted result set
        for xd, p in cands:
            if xd < R[-1][0] or len(R) < ef:
                heap_push(C, (-xd, p))
                log.append((xd, p))
                pos = sum(1 for e in R if e[0] <= xd)
                R.insert(pos, [xd, p, False, False])
                if len(R) > ef:
                    if R[-1][0] == R[-2][0]:
                        for e in R:
                            if e[0] == R[-1][0]:
                                e[3] = True
                    R.pop()

    while C is None:
        cj = next((j for j, e in enumerate(R) if not e[2]), None)
        if cj is None:
            break
        R[cj][2] = True
        c = R[cj][1]
        log.append(EXPAND)
        lst = g.nbrs(c, 0)
        for b0 in range(0, len(lst), batch):
            fresh = [p for p in lst[b0:b0 + batch] if p not in visited]
            visited.update(fresh)
            de = [(g.d(q, p), p) for p in fresh]
            worst = R[-1][0]
            cand = [x for x in de if len(R) < ef or x[0] < worst]
            for i, (xd, p) in enumerate(cand):
                if xd < worst or len(R) < ef:
                    log.append((xd, p))
                    tie = any(e[0] == xd for e in R)
                    pos = sum(1 for e in R if e[0] <= xd)
                    R.insert(pos, [xd, p, False, False])
                    if len(R) > ef:
                        R.pop()       # (no equal pair so far: the evicted entry was the unique farthest)
                    if tie:
                        C = []
                        heap_push(C, (-dcur, pivot))
                        for op in log:
                            if op is EXPAND:
                                heap_pop(C)
                            else:
                                heap_push(C, (-op[0], op[1]))
                        offer(cand[i + 1:])
                        for b1 in range(b0 + batch, len(lst), batch):
                            fresh = [p2 for p2 in lst[b1:b1 + batch] if p2 not in visited]
                            visited.update(fresh)
                            w = R[-1][0]
                            offer([x for x in [(g.d(q, p2), p2) for p2 in fresh] if len(R) < ef or x[0] < w])
                        break
                    worst = R[-1][0]
            if C is not None:
                break
    if C is not None:
        while C:
            ce = heap_pop(C)
            if -ce[0] > R[-1][0]:
                break
            lst = g.nbrs(ce[1], 0)
            for b0 in range(0, len(lst), batch):
                fresh = [p for p in lst[b0:b0 + batch] if p not in visited]
                visited.update(fresh)
                w = R[-1][0]
                offer([x for x in [(g.d(q, p), p) for p in fresh] if len(R) < ef or x[0] < w])
    kk = min(k, ef, len(R))
    need = any(e[3] for e in R[:kk]) or any(R[j][0] == R[j + 1][0] for j in range(kk) if j + 1 < len(R))
    if not need:
        return [e[1][0] for e in R[:kk]], C is not None, False
    RR = []
    heap_push(RR, (dcur, pivot))
    for op in log:
        if op is not EXPAND:
            heap_push(RR, op)
            if len(RR) > ef:
                heap_pop(RR)
    return [e[1][0] for e in heap_into_sorted(RR)[:kk]], C is not None, True


@pytest.mark.parametrize("kind,dist,ef,m", [("grid", "DistL2", 16, 8), ("duplicates", "DistL2", 24, 8), ("grid", "DistL1", 40, 40),
                                             ("uniform", "DistL2", 32, 12), ("sparse-ties", "DistL2", 24, 10)])
def test_end_state_design_equals_the_literal_search(native, oracle, tmp_path, kind, dist, ef, m):
    rng = np.random.default_rng(29)
    n, d = 1200, 6
    if kind == "duplicates":
        base = rng.random((n // 2, d), dtype=np.float32)
        X = np.concatenate([base, base])[rng.permutation(n)]
    elif kind == "grid":
        X = rng.integers(0, 4, (n, d)).astype(np.float32)
    elif kind == "sparse-ties":          # mostly distinct distances, a few exact duplicates: like natural data
        X = rng.random((n, d), dtype=np.float32)
        X[rng.choice(n, 60, replace=False)] = X[rng.choice(n, 60, replace=False)]
    else:
        X = rng.random((n, d), dtype=np.float32)
    X = np.ascontiguousarray(X)
    o = oracle.OracleHnsw(m, n, 16, 60, dist)
    o.insert_batch(X)
    o.file_dump(tmp_path, "e")
    h = native.HnswIo(tmp_path, "e").load_hnsw(dist)
    g = Graph(h, X, dist, oracle)
    handed = rebuilt = 0
    for qi in range(50):
        q = rng.integers(0, 4, d).astype(np.float32) if kind == "grid" else rng.random(d, dtype=np.float32)
        ids, was, rep = search_end_state(g, q, 10, ef)
        handed += was
        rebuilt += rep
        ref_ids, _, _, _ = o.search(q, 10, ef)
        assert ids == ref_ids.tolist(), f"query {qi} (handed over: {was}, result heap rebuilt: {rep})"
    print(f"{kind} {dist} ef={ef}: {handed} of 50 queries handed over to the literal candidate heap, {rebuilt} rebuilt return_points")
