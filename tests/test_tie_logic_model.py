"""The search kernel's way of taking the reference's decisions (hnswlib-rs_amd/csrc/search_kernels.inc, header of hnsw_search_kernel;
DESIGN.md 6): ONE sorted array with EXPANDED flags stands for both BinaryHeaps of `search_layer` (src/hnsw.rs:922-1064) for as long as
values decide; a pop that equal distances make ambiguous -- (A) two equal nearest candidates, (B) a "ghost" evicted between equal
farthest entries -- is taken from the literal candidate heap, brought up to date from the log of heap operations; and (C) when equal
distances reach the first k answers, the literal return_points is rebuilt from the same log and sorted the reference's way.

Here that ALGORITHM is restated in Python on top of the lane-level result set of tests/test_result_set_emulation.py (r_insert,
merge_list) and std's BinaryHeap of tests/test_oracle.py, and fuzzed against a plain transcription of the reference's loop on
thousands of small random graphs whose distances are drawn from a handful of values (every search is full of ties): the answers
-- ids in order -- must be identical, with the batch accept path on and off, lists longer than one batch of 64 ids included."""
import random

import pytest

from test_oracle import _StdBinaryHeap
from test_result_set_emulation import DevR, LANES, ballot

EXPANDED = 1 << 31


def reference_search(neigh, dist, entry, ef, k):
    """search_layer with filter = None, then into_sorted_vec + truncate (src/hnsw.rs:1544-1578)"""
    cand, ret = _StdBinaryHeap(), _StdBinaryHeap()
    visited = {entry}
    cand.push((-dist[entry], entry))
    ret.push((dist[entry], entry))
    while cand.d:
        c = cand.pop()
        f = ret.d[0]
        if -c[0] > f[0]:
            break
        for e in neigh[c[1]]:
            if e in visited:
                continue
            visited.add(e)
            f = ret.d[0]
            if dist[e] < f[0] or len(ret.d) < ef:
                cand.push((-dist[e], e))
                ret.push((dist[e], e))
                if len(ret.d) > ef:
                    ret.pop()
    return [i for _, i in ret.into_sorted_vec()[:k]]


def device_search(neigh, dist, entry, ef, k, merge):
    S = 1 if ef <= 64 else 2 if ef <= 128 else 4
    R = DevR(S, ef, [(dist[entry], entry)])
    visited = {entry}
    log = [("push", dist[entry], entry)]                    # entry 0 of the log: the entry point
    C, replayed = _StdBinaryHeap(), 0
    ghost, ghost_w, taint_w = 0, None, None
    tie_any = literal_used = False

    def arr():
        return R.array()

    while True:
        a = arr()
        cj = next((j for j, (_, i) in enumerate(a) if not i & EXPANDED), None)
        amb = False
        if tie_any:
            if cj is not None:
                dcj = a[cj][0]
                n_equal = sum(1 for d, i in a if d == dcj and not i & EXPANDED)
                amb = n_equal >= 2 or (ghost > 0 and dcj == ghost_w)
            else:
                amb = ghost > 0
        popped = False
        if amb:
            for op in log[replayed:]:                       # the lazy heap: brought up to date from the log
                if op[0] == "push":
                    C.push((-op[1], op[2]))
                else:
                    C.pop()
            replayed = len(log)
            if not C.d:
                break                                       # candidate_points is empty (:969)
            ce = C.pop()
            literal_used = popped = True
            log.append(("pop",))
            replayed = len(log)                             # the heap has already performed this pop
            if -ce[0] > a[-1][0]:
                break                                       # :973-993
            c = ce[1]
            found = False
            for j, (d, i) in enumerate(a):
                if i == c:                                  # (an expanded entry carries the flag: no match)
                    R.ri[j >> 6][j & 63] |= EXPANDED
                    found = True
            if not found and ghost > 0:
                ghost -= 1
        if not popped:
            if cj is None:
                break
            c = a[cj][1]
            R.ri[cj >> 6][cj & 63] |= EXPANDED
            log.append(("pop",))
        lst = neigh[c]
        for b0 in range(0, len(lst), 64):                   # a neighbour list is scanned in batches of 64 ids
            fresh = [e for e in lst[b0:b0 + 64] if e not in visited]
            visited.update(fresh)
            nf = len(fresh)
            if nf == 0:
                continue
            de = [dist[fresh[l]] if l < nf else 9e9 for l in LANES]
            idc = [fresh[l] if l < nf else -1 for l in LANES]
            worst = arr()[-1][0]
            cand = ballot([l < nf and (R.len < ef or de[l] < worst) for l in LANES])
            if merge and R.len == ef and cand:
                ok, tie_here, taken = R.merge_list(cand, de, idc)
                if ok:
                    if taken:
                        tie_any = tie_any or tie_here
                        ghost = 0
                        log.extend(("push", de[l], idc[l]) for l in LANES if (taken >> l) & 1)
                    cand = 0
            while cand:
                jl = (cand & -cand).bit_length() - 1
                cand &= cand - 1
                xd = de[jl]
                if xd < worst or R.len < ef:
                    log.append(("push", xd, idc[jl]))
                    was_full = R.len == ef
                    last_if = arr()[ef - 1][1] if was_full else 0
                    if R.r_insert(xd, idc[jl]):
                        tie_any = True
                    nw = arr()[-1][0]
                    if was_full:
                        if nw == worst:
                            taint_w = nw
                            if not last_if & EXPANDED:
                                ghost += 1
                                ghost_w = nw
                        else:
                            ghost = 0
                    worst = nw
    a = arr()
    kk = min(k, len(a))
    ambiguous = kk > 0 and (any(j + 1 < len(a) and a[j][0] == a[j + 1][0] for j in range(kk)) or a[kk - 1][0] == taint_w)
    if tie_any and ambiguous:                               # (C): the literal return_points from the log
        ret = _StdBinaryHeap()
        for op in log:
            if op[0] == "push":
                ret.push((op[1], op[2]))
                if len(ret.d) > ef:
                    ret.pop()
        out = ret.into_sorted_vec()
        return [i for _, i in out[:min(len(out), ef)][:k]], literal_used or True
    return [i & ~EXPANDED for _, i in a[:k]], literal_used


def random_case(rnd):
    n = rnd.choice([6, 12, 30, 80, 200])
    nvals = rnd.choice([1, 2, 3, 5, 9, 40])
    vals = [float(v) for v in rnd.sample(range(1, 500), nvals)]
    dist = [rnd.choice(vals) for _ in range(n)]
    if rnd.random() < 0.3:                                  # some searches with mostly distinct distances and a few equal pairs
        dist = [float(i) for i in rnd.sample(range(1, 10 * n), n)]
        for _ in range(rnd.randint(1, max(1, n // 4))):
            dist[rnd.randrange(n)] = dist[rnd.randrange(n)]
    maxdeg = rnd.choice([2, 4, 8, 16, 32, 48, 100]) if n > 100 else rnd.choice([2, 4, 8, 16, 32])
    neigh = []
    for v in range(n):
        deg = rnd.randint(0, min(maxdeg, n - 1))
        neigh.append(rnd.sample([u for u in range(n) if u != v], deg))
    ef = rnd.choice([1, 2, 3, 5, 8, 16, 24, 64, 65, 100])
    k = rnd.choice([1, 2, 3, 5, 10, ef])
    ef = max(ef, k)                                         # :1531
    if ef > 256:
        ef = k = 100
    return neigh, dist, rnd.randrange(n), ef, k


@pytest.mark.parametrize("seed", range(8))
def test_sorted_array_with_lazy_literal_heaps_answers_like_the_two_binary_heaps(seed):
    rnd = random.Random(9000 + seed)
    literal = 0
    for rep in range(700):
        neigh, dist, entry, ef, k = random_case(rnd)
        want = reference_search(neigh, dist, entry, ef, k)
        for merge in (True, False):
            got, used = device_search(neigh, dist, entry, ef, k, merge)
            assert got == want, (seed, rep, merge, ef, k, entry, dist, neigh)
            literal += bool(used)
    assert literal > 100                                     # the literal paths were really taken
