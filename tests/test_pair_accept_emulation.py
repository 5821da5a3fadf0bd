"""The accept rule of hnsw_search_pair_kernel (csrc/search_pair.inc), emulated lane for lane in Python against the reference's
sequential rule (src/hnsw.rs:1028-1053: a neighbour enters return_points when the heap holds fewer than ef entries or it is nearer
than the farthest entry, which is then evicted).  The kernel decides a whole round of <= 16 candidates at once:
    accepted_j  <=>  cr_j + cb_j < ef,   cr_j = entries of R not farther than d_j,  cb_j = EARLIER candidates not farther than d_j
(all earlier candidates, not only the accepted ones), an accepted candidate lands at cr_j + its rank among the accepted, an entry of
R moves up by the accepted candidates whose cr is <= its index.  R is always ef entries long (+inf sentinels).  Runs on the CPU."""
import bisect
import math

import numpy as np
import pytest

INF = math.inf


def reference_round(R, ef, cands):
    """R: list of (d, id) sorted by (d, arrival); sequential insertion, equal distances in arrival order, farthest evicted."""
    R = list(R)
    for d, i in cands:
        if len(R) < ef or d < R[-1][0]:
            pos = bisect.bisect_right([e[0] for e in R], d)
            R.insert(pos, (d, i))
            if len(R) > ef:
                R.pop()
    return R


def pair_round(R, ef, cands):
    """What one half-wave does (sentinel-padded array, vector steps only)."""
    keys = [e[0] for e in R] + [INF] * (ef - len(R))
    ents = list(R) + [(INF, None)] * (ef - len(R))
    worst = keys[ef - 1]
    passing = [(d, i) for d, i in cands if d < worst]          # `pass` lanes, compacted in list order
    if not passing:
        return [e for e in ents if e[1] is not None]
    p2 = 1
    while p2 < ef:
        p2 *= 2
    cr = []
    for d, _ in passing:                                         # the binary search of every candidate lane in the LDS mirror
        pos, step = 0, p2 >> 1
        while step:
            idx = pos + step - 1
            if idx < ef and keys[idx] <= d:
                pos += step
            step >>= 1
        cr.append(pos)
    cb = [sum(1 for t in range(j) if passing[t][0] <= passing[j][0]) for j in range(len(passing))]   # DPP row_shr steps
    acc = [cr[j] + cb[j] < ef for j in range(len(passing))]
    aft = [sum(1 for t in range(j + 1, len(passing)) if acc[t] and passing[t][0] < passing[j][0]) for j in range(len(passing))]
    out = [None] * ef
    acr = [cr[j] for j in range(len(passing)) if acc[j]]
    for e in range(ef):                                          # entries of R: shifted by the accepted candidates nearer than them
        sh = sum(1 for c in acr if c <= e)
        if e + sh < ef:
            assert out[e + sh] is None
            out[e + sh] = ents[e]
    for j in range(len(passing)):
        if acc[j]:
            fpos = cr[j] + cb[j] + aft[j]
            if fpos < ef:
                assert out[fpos] is None
                out[fpos] = passing[j]
    assert all(o is not None for o in out)
    return [e for e in out if e[1] is not None]


@pytest.mark.parametrize("seed", range(40))
def test_pair_accept_rule_equals_the_sequential_rule(seed):
    rng = np.random.default_rng(seed)
    ef = int(rng.choice([1, 2, 3, 5, 8, 16, 24, 48, 64, 100, 128]))
    levels = int(rng.choice([3, 8, 1000]))                       # few levels: equal distances everywhere
    R, next_id = [], 0
    for _ in range(60):
        n = int(rng.integers(1, 17))
        cands = []
        for _ in range(n):
            cands.append((float(rng.integers(0, levels)) / 7.0, next_id))
            next_id += 1
        want = reference_round(R, ef, cands)
        got = pair_round(R, ef, cands)
        assert got == want, (ef, R, cands)
        R = want


# ---------------------------------------------------------------------------------------------------------------------------
# The whole search of a half-wave, with the three places where it hands a query on (search_pair.inc): (A) a pop whose entry has an
# equal, still unexpanded successor; (B) equal entries on both sides of the cut at ef after a rebuild; (C) equal distances inside
# the first k answers or across their end.  Claim: a search that meets none of them answers like the reference's two BinaryHeaps
# (std's heap order included) -- fuzzed on small random graphs full of equal distances against the literal transcription of
# tests/test_tie_logic_model.py.
# ---------------------------------------------------------------------------------------------------------------------------
def pair_search(neigh, dist, entry, ef, k, deg_rounds=16):
    """Returns (answer ids, handed_on)."""
    keys = [dist[entry]] + [INF] * (ef - 1)
    ids = [entry] + [None] * (ef - 1)
    expanded = [False] + [True] * (ef - 1)          # sentinels never get popped
    length = 1
    visited = {entry}
    while True:
        un = [j for j in range(ef) if not expanded[j]]
        if not un:
            break
        cj = un[0]
        if len(un) > 1 and keys[un[1]] == keys[cj]:
            return None, True                        # (A)
        expanded[cj] = True
        fresh = []
        for e in neigh[ids[cj]]:
            if e not in visited:
                visited.add(e)
                fresh.append((dist[e], e))
        for s0 in range(0, len(fresh), deg_rounds):  # rounds of <= 16 rows, the accept rule per round
            cands = fresh[s0:s0 + deg_rounds]
            worst = keys[ef - 1]
            passing = [c for c in cands if c[0] < worst]
            if not passing:
                continue
            cr = [bisect.bisect_right(keys, d) for d, _ in passing]     # (keys: sorted, +inf sentinels behind the entries)
            cb = [sum(1 for t in range(j) if passing[t][0] <= passing[j][0]) for j in range(len(passing))]
            acc = [cr[j] + cb[j] < ef for j in range(len(passing))]
            aft = [sum(1 for t in range(j + 1, len(passing)) if acc[t] and passing[t][0] < passing[j][0]) for j in range(len(passing))]
            merged = [None] * (ef + len(passing))
            acr = [cr[j] for j in range(len(passing)) if acc[j]]
            for e in range(ef):
                merged[e + sum(1 for c in acr if c <= e)] = (keys[e], ids[e], expanded[e])
            for j in range(len(passing)):
                if acc[j]:
                    merged[cr[j] + cb[j] + aft[j]] = (passing[j][0], passing[j][1], False)
            na = sum(acc)
            assert all(m is not None for m in merged[:ef + na])
            dropped = merged[ef]
            keys = [m[0] for m in merged[:ef]]
            ids = [m[1] for m in merged[:ef]]
            expanded = [m[2] for m in merged[:ef]]
            length = min(ef, length + na)
            if dropped[1] is not None and dropped[0] == keys[ef - 1]:
                return None, True                    # (B): the cut at ef runs through equal entries
    kk = min(k, length)
    for j in range(kk):
        if j + 1 < length and keys[j] == keys[j + 1]:
            return None, True                        # (C)
    return ids[:kk], False


@pytest.mark.parametrize("seed", range(6))
def test_pair_search_that_hands_nothing_on_answers_like_the_two_binary_heaps(seed):
    import random
    from test_tie_logic_model import random_case, reference_search
    rnd = random.Random(4200 + seed)
    answered = handed = 0
    for rep in range(700):
        neigh, dist, entry, ef, k = random_case(rnd)
        if ef > 128:
            continue
        want = reference_search(neigh, dist, entry, ef, k)
        got, on = pair_search(neigh, dist, entry, ef, k)
        if on:
            handed += 1
            continue
        answered += 1
        assert got == want, (seed, rep, ef, k, entry, dist, neigh)
    assert answered > 100 and handed > 100           # both outcomes were really exercised
