"""The result set of the search kernel (hnswlib-rs_amd/csrc/search_kernels.inc): ONE array sorted ascending by distance, entry j in
slot j / 64 of lane j % 64, equals in arrival order.  Restated lane by lane in Python:
* r_insert -- position by ballot, the shift by one entry across lanes and slots, the eviction past ef - 1, the tie report;
* merge_list -- the accept rule of the reference (src/hnsw.rs:1028-1053: a neighbour enters when it is nearer than the farthest entry
  AT ITS TURN) applied to a whole neighbour list at once through a scatter buffer, refused (nothing changed) when equal entries end up
  on both sides of the cut at ef;
and checked against the plain sequential model (insert after the equals, drop the last): whenever merge_list accepts, the array and the
tie report equal those of the neighbours taken one at a time, and the bookkeeping it skips (ghosts, taint_w) would have ended without
meaning: no entry of an ambiguous eviction's distance is left; when it refuses, nothing was touched and the sequential pass really
met equal farthest entries.  Distances are drawn from a handful of values: ties everywhere."""
import random

import pytest

LANES = range(64)


def ballot(pred):
    m = 0
    for l in LANES:
        if pred[l]:
            m |= 1 << l
    return m


def popc(x):
    return bin(x).count("1")


class DevR:
    def __init__(self, S, ef, entries):
        self.S, self.ef = S, ef
        self.rd = [[9e9] * 64 for _ in range(S)]      # lanes beyond len hold leftovers: anything
        self.ri = [[-7] * 64 for _ in range(S)]
        self.len = len(entries)
        for j, (d, i) in enumerate(entries):
            self.rd[j >> 6][j & 63], self.ri[j >> 6][j & 63] = d, i

    def array(self):
        return [(self.rd[j >> 6][j & 63], self.ri[j >> 6][j & 63]) for j in range(self.len)]

    def r_insert(self, xd, xi):
        S, ln = self.S, self.len
        pos, tie = 0, False
        for s in range(S):
            pos += popc(ballot([64 * s + l < ln and self.rd[s][l] <= xd for l in LANES]))
            tie = tie or ballot([64 * s + l < ln and self.rd[s][l] == xd for l in LANES]) != 0
        for s in range(S - 1, -1, -1):
            # entry j - 1 -> j: wave_shr:1 inside a slot (lane 0 keeps `old`), lane 63 of the previous slot into lane 0
            od0, oi0 = (self.rd[s - 1][63], self.ri[s - 1][63]) if s > 0 else (self.rd[s][0], self.ri[s][0])
            pd = [od0] + self.rd[s][:63]
            pi = [oi0] + self.ri[s][:63]
            for l in LANES:
                j = 64 * s + l
                if j > pos:
                    self.rd[s][l], self.ri[s][l] = pd[l], pi[l]
                elif j == pos:
                    self.rd[s][l], self.ri[s][l] = xd, xi
        self.len = min(self.ef, ln + 1)
        return tie

    def merge_list(self, cand, de, idc):
        """returns (accepted, tie_here, lanes taken); the set must be full"""
        S, ef = self.S, self.ef
        assert self.len == ef
        slotmask = []
        for s in range(S):
            lo = 64 * s
            slotmask.append((1 << 64) - 1 if ef >= lo + 64 else ((1 << (ef - lo)) - 1 if ef > lo else 0))
        acc, tie_here = 0, False
        shift = [[0] * 64 for _ in range(S)]
        pos_c = [0] * 64
        while cand:
            j = (cand & -cand).bit_length() - 1
            cand &= cand - 1
            xd = de[j]
            cr, eqm = 0, 0
            for s in range(S):
                cr += popc(ballot([self.rd[s][l] <= xd for l in LANES]) & slotmask[s])
                eqm |= ballot([self.rd[s][l] == xd for l in LANES]) & slotmask[s]
            cb = popc(ballot([de[l] <= xd for l in LANES]) & acc)
            if cr + cb >= ef:
                continue
            if eqm | (ballot([de[l] == xd for l in LANES]) & acc):
                tie_here = True
            for s in range(S):
                for l in LANES:
                    shift[s][l] += 1 if xd < self.rd[s][l] else 0
            for l in LANES:
                pos_c[l] += 1 if xd < de[l] else 0
            pos_c[j] = cr + cb
            acc |= 1 << j
        if acc == 0:
            return True, False, 0
        buf = {}
        for s in range(S):
            for l in LANES:
                if (slotmask[s] >> l) & 1:
                    k = 64 * s + l + shift[s][l]
                    assert k not in buf
                    buf[k] = (self.rd[s][l], self.ri[s][l])
        for l in LANES:
            if (acc >> l) & 1:
                assert pos_c[l] not in buf
                buf[pos_c[l]] = (de[l], idc[l])
        assert sorted(buf) == list(range(ef + popc(acc)))      # the scatter is a permutation of a prefix of the buffer
        out0 = buf[ef]
        w_new = buf[ef - 1][0]
        if out0[0] == w_new:
            return False, tie_here, acc
        for s in range(S):
            for l in LANES:
                if (slotmask[s] >> l) & 1:
                    self.rd[s][l], self.ri[s][l] = buf[64 * s + l]
        return True, tie_here, acc


def sequential(entries, ef, cands):
    """the reference's rule, one neighbour at a time, on a plain list, with the kernel's bookkeeping of its one-at-a-time path
    (ghost: unexpanded entries evicted while an equal one stayed; taint_w: the distance of the last eviction between equal farthest
    entries; both only mean something while entries of that distance are still in the set).  Returns (array, tie met, taken lanes,
    ghost, taint_w)."""
    r = list(entries)
    tie = False
    taken, ghost, taint_w = 0, 0, None
    for lane, (xd, xi) in cands:
        worst = r[-1][0]
        if len(r) == ef and not xd < worst:
            continue
        pos = sum(1 for d, _ in r if d <= xd)
        tie = tie or any(d == xd for d, _ in r)
        was_full = len(r) == ef
        r.insert(pos, (xd, xi))
        del r[ef:]
        if was_full:
            if r[-1][0] == worst:
                taint_w = worst
                ghost += 1          # (none of the test's entries is expanded)
            else:
                ghost = 0
        taken |= 1 << lane
    return r, tie, taken, ghost, taint_w


@pytest.mark.parametrize("S,ef", [(1, 1), (1, 2), (1, 17), (1, 63), (1, 64), (2, 65), (2, 100), (2, 128), (4, 129), (4, 200), (4, 256)])
def test_r_insert_keeps_the_array_the_reference_way(S, ef):
    rnd = random.Random(S * 1000 + ef)
    for rep in range(30):
        vals = [float(v) for v in rnd.sample(range(1, 40), rnd.choice([3, 8, 30]))]
        dev, model, nid = DevR(S, ef, []), [], 0
        for _ in range(3 * ef + 10):
            xd = rnd.choice(vals)
            if len(model) == ef and not xd < model[-1][0]:      # (the caller's test: only nearer than the farthest enters a full set)
                continue
            tie_model = any(d == xd for d, _ in model)
            model.insert(sum(1 for d, _ in model if d <= xd), (xd, nid))
            del model[ef:]
            assert dev.r_insert(xd, nid) == tie_model
            nid += 1
            assert dev.array() == model


@pytest.mark.parametrize("S,ef", [(1, 1), (1, 2), (1, 10), (1, 64), (2, 65), (2, 128), (4, 200), (4, 256)])
def test_merge_list_equals_the_neighbours_taken_one_at_a_time(S, ef):
    rnd = random.Random(S * 7919 + ef)
    accepted = refused = 0
    for rep in range(400):
        nvals = rnd.choice([2, 4, 12, 60, 2000])                        # (2 000 values: ties are the exception, as in a real search)
        vals = [float(v) for v in rnd.sample(range(1, 100 if nvals <= 60 else 100000), nvals)]
        entries = sorted(((rnd.choice(vals), 10_000 + i) for i in range(ef)), key=lambda e: e[0])   # equals: ids ascending = arrival order
        nf = rnd.randint(1, 64)
        de = [rnd.choice(vals + [0.5, 200000.0]) for _ in LANES]
        idc = list(range(64))
        worst = entries[-1][0]
        cand = ballot([l < nf and de[l] < worst for l in LANES])
        dev = DevR(S, ef, entries)
        before = dev.array()
        ok, tie_here, taken = dev.merge_list(cand, de, idc)
        want, tie_seq, taken_seq, ghost, taint_w = sequential(entries, ef, [(l, (de[l], idc[l])) for l in LANES if (cand >> l) & 1])
        if ok:
            accepted += 1
            assert dev.array() == want, (rep, entries, de[:nf])
            assert taken == taken_seq and tie_here == tie_seq
            # the batch path resets the ghosts and leaves taint_w alone: right, because whatever eviction between equal farthest
            # entries the one-at-a-time pass met on the way involved only entries that have all left by the end
            if taken:
                assert ghost == 0
            assert taint_w is None or taint_w > want[-1][0]
        else:
            refused += 1
            assert dev.array() == before  # nothing changed: the caller takes the neighbours one at a time
            assert taint_w is not None    # and the one-at-a-time pass does meet equal farthest entries
    assert accepted > 50
    if ef > 2:
        assert refused > 0
