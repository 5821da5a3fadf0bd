"""The lane-parallel heap primitives of the literal BinaryHeap emulation (hnswlib-rs_amd/csrc/search_kernels.inc: heap_push,
heap_chase + heap_pop, heap_pop3, heap_sift_down_range), restated lane by lane in Python -- 64 lanes, ballots, the loads of an
operation before its stores -- and fuzzed against the transcription of Rust std's BinaryHeap in tests/test_oracle.py: random and
tie-saturated keys, heaps from 1 entry to several thousand (three 5-level rounds of heap_pop3), pushes and pops interleaved, and
into_sorted_vec through heap_sift_down_range.  The device code cannot run here; its ALGORITHM can, and one of its siblings
(heap_push_batch16, tests/test_heap_batch_push.py) had a bug that only a soak on the GPU had found."""
import random

import pytest

from test_oracle import _StdBinaryHeap

LANES = range(64)


def ballot(pred):
    m = 0
    for l in LANES:
        if pred[l]:
            m |= 1 << l
    return m


def ctz(x):
    return (x & -x).bit_length() - 1


def highest(x):
    return x.bit_length() - 1


def key(e):
    return e[0]


ZERO = (0.0, -1)  # what an invalid lane holds (hent 0: key +0.0)


def dev_push(data, item):
    pos = len(data)
    data.append(None)
    if pos == 0 or key(item) <= key(data[(pos - 1) >> 1]):
        data[pos] = item
        return
    p1 = [((pos + 1) >> l) if l < 32 else 0 for l in LANES]
    anc = [l >= 1 and p1[l] >= 1 for l in LANES]
    e = [data[p1[l] - 1] if anc[l] else ZERO for l in LANES]
    am = ballot(anc)
    stop = ballot([anc[l] and key(item) <= key(e[l]) for l in LANES])
    depth = bin(am).count("1")
    moved = ctz(stop) - 1 if stop else depth
    stores = []
    for l in LANES:
        if 1 <= l <= moved:
            stores.append((((pos + 1) >> (l - 1)) - 1, e[l]))
    stores.append((((pos + 1) >> moved) - 1, item))
    for i, v in stores:
        data[i] = v


def dev_chase(data, end):
    """returns m, my_pos[], my_ent[]"""
    m, p = 0, 0
    my_pos, my_ent = [0] * 64, [ZERO] * 64
    while True:
        idx, e, valid = [0] * 64, [ZERO] * 64, [False] * 64
        for l in LANES:
            L1 = l + 1
            t = L1.bit_length() - 1
            i64 = ((p + 1) << t) - 1 + (L1 - (1 << t))
            valid[l] = l < 63 and i64 < end
            idx[l] = i64 & 0xFFFFFFFF
            if valid[l]:
                e[l] = data[idx[l]]
        vm = ballot(valid)
        cur, bottom = 0, False
        for _step in range(5):
            lch, rch = 2 * cur + 1, 2 * cur + 2
            if not (vm >> lch) & 1:
                bottom = True
                break
            nxt = lch
            if (vm >> rch) & 1:
                nxt = rch if key(e[lch]) <= key(e[rch]) else lch
            m += 1
            my_pos[m], my_ent[m] = idx[nxt], e[nxt]
            cur = nxt
        if bottom:
            break
        p = idx[cur]
    return m, my_pos, my_ent


def dev_pop(data):
    last = data.pop()
    if not data:
        return last
    root = data[0]
    m, my_pos, my_ent = dev_chase(data, len(data))
    le = ballot([1 <= l <= m and key(last) <= key(my_ent[l]) for l in LANES])
    jstar = highest(le) if le else 0
    stores = []
    for l in LANES:
        if 1 <= l <= jstar:
            stores.append((my_pos[l - 1], my_ent[l]))
        if l == jstar:
            stores.append((my_pos[l], last))
    for i, v in stores:
        data[i] = v
    return root


def dev_pop3(data):
    last = data.pop()
    if not data:
        return last
    end = len(data)
    assert end < 32768
    t = [(l + 1).bit_length() - 1 for l in LANES]
    off = [(l + 1) - (1 << t[l]) for l in LANES]
    idx = [[0] * 64 for _ in range(3)]
    ent = [[ZERO] * 64 for _ in range(3)]
    pathm, lem = [0, 0, 0], [0, 0, 0]
    root, p, more = None, 0, True
    for r in range(3):
        if not more:
            continue
        valid = [False] * 64
        for l in LANES:
            i64 = ((p + 1) << t[l]) - 1 + off[l]
            valid[l] = l < 63 and i64 < end
            idx[r][l] = i64 & 0xFFFFFFFF
            ent[r][l] = data[idx[r][l]] if valid[l] else ZERO
        if r == 0:
            root = ent[0][0]
        vm = ballot(valid)
        hl, pr = [False] * 64, [False] * 64
        for l in LANES:
            cl = 2 * l + 1
            kl, kr = key(ent[r][cl & 63]), key(ent[r][(cl + 1) & 63])
            hl[l] = l <= 30 and (vm >> cl) & 1 != 0
            pr[l] = l <= 30 and (vm >> (cl + 1)) & 1 != 0 and kl <= kr
        hm, pm = ballot(hl), ballot(pr)
        cur, bits, more = 0, 0, False
        for step in range(5):
            if not (hm >> cur) & 1:
                break
            cur = 2 * cur + 1 + ((pm >> cur) & 1)
            bits |= 1 << cur
            if step == 4:
                more = True
        pathm[r] = bits
        lem[r] = ballot([(bits >> l) & 1 != 0 and key(last) <= key(ent[r][l]) for l in LANES])
        if more:
            p = idx[r][cur]
    rstar = -1
    for r in range(3):
        if lem[r]:
            rstar = r
    stores = []
    if rstar < 0:
        stores.append((0, last))
    else:
        for r in range(3):
            if r <= rstar:
                lstar = highest(lem[r]) if r == rstar else 63
                for l in LANES:
                    if (pathm[r] >> l) & 1 and l <= lstar:
                        stores.append(((idx[r][l] - 1) >> 1, ent[r][l]))
                    if r == rstar and l == lstar:
                        stores.append((idx[r][l], last))
    for i, v in stores:
        data[i] = v
    return root


def dev_sift_down_range(data, end):
    elt = data[0]
    m, my_pos, my_ent = dev_chase(data, end)
    ge = ballot([1 <= l <= m and key(elt) >= key(my_ent[l]) for l in LANES])
    nshift = ctz(ge) - 1 if ge else m
    stores = []
    for l in LANES:
        if 1 <= l <= nshift:
            stores.append((my_pos[l - 1], my_ent[l]))
        if l == nshift:
            stores.append((my_pos[l], elt))
    for i, v in stores:
        data[i] = v


def dev_into_sorted_vec(data):
    end = len(data)
    while end > 1:
        end -= 1
        data[0], data[end] = data[end], data[0]
        dev_sift_down_range(data, end)
    return data


def _keys(rnd, ties):
    return (lambda: float(rnd.choice([0, 1, 2, 3, 5, 8, 9]))) if ties else rnd.random


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("pop", [dev_pop, dev_pop3])
def test_lane_parallel_push_and_pop_equal_std(ties, pop):
    rnd = random.Random(101 + ties)
    for rep in range(25):
        k = _keys(rnd, ties)
        a, data, t = _StdBinaryHeap(), [], 0
        grow = rnd.choice([0.55, 0.7, 0.9])        # random walk of the size: small heaps, and heaps of thousands
        for _ in range(rnd.choice([60, 400, 5000])):
            if data and rnd.random() > grow:
                assert a.pop() == pop(data)
            else:
                it = (k(), t)
                t += 1
                a.push(it)
                dev_push(data, it)
            assert a.d == data
        while data:                                  # and down to the last entry
            assert a.pop() == pop(data)
            assert a.d == data


@pytest.mark.parametrize("ties", [False, True])
def test_three_rounds_of_pop3_on_deep_heaps(ties):
    """heaps of 1 023 .. 20 000 entries: the greater-child walk crosses two round boundaries (5 + 5 + up to 5 levels)"""
    rnd = random.Random(7 + ties)
    k = _keys(rnd, ties)
    for n in (1023, 1024, 2047, 2048, 4000, 20000):
        a, t = _StdBinaryHeap(), 0
        for _ in range(n):
            a.push((k(), t))
            t += 1
        data = list(a.d)
        for _ in range(300):
            assert a.pop() == dev_pop3(data)
            assert a.d == data
            it = (k(), t)
            t += 1
            a.push(it)
            dev_push(data, it)
            assert a.d == data


@pytest.mark.parametrize("ties", [False, True])
def test_into_sorted_vec_through_sift_down_range(ties):
    rnd = random.Random(31 + ties)
    k = _keys(rnd, ties)
    for n in list(range(0, 20)) + [63, 64, 65, 200, 513, 1300]:
        a = _StdBinaryHeap()
        for t in range(n):
            a.push((k(), t))
        data = list(a.d)
        assert dev_into_sorted_vec(data) == a.into_sorted_vec()


# ------------------------------------------------------------------------------------------------- RegHeap: return_points in VGPRs
class DevRegHeap:
    """RegHeap<NS> (search_kernels.inc): entry i sits in slot i // 64 of lane i % 64; push, pop, the fused push + pop on a full
    heap, sift_down_range -- with the slot arithmetic of chase() (children of slot k in slots 2k and 2k + 1, the right child of
    a slot's last node in slot 2k + 2) restated lane by lane."""

    def __init__(self, ns):
        self.ns = ns
        self.cap = 64 * ns
        self.kc = 1 if ns == 1 else ns // 2
        self.s = [[ZERO] * 64 for _ in range(ns)]

    def get(self, i):
        return self.s[i >> 6][i & 63] if (i >> 6) < self.ns else ZERO

    def set(self, i, v):
        if (i >> 6) < self.ns:
            self.s[i >> 6][i & 63] = v

    def push(self, ln, item):
        pos = ln
        while pos > 0:
            parent = (pos - 1) >> 1
            pe = self.get(parent)
            if key(item) <= key(pe):
                break
            self.set(pos, pe)
            pos = parent
        self.set(pos, item)
        return ln + 1

    def chase(self, end):
        ns, kc, s = self.ns, self.kc, self.s
        chosen = [[ZERO] * 64 for _ in range(kc)]
        hasl, prefr, pathm = [0] * kc, [0] * kc, [0] * kc
        for k in range(kc):
            hl, pr = [False] * 64, [False] * 64
            for lane in LANES:
                i = 64 * k + lane
                cl, cr = 2 * i + 1, 2 * i + 2
                if ns == 1:
                    L, R = s[0][cl & 63], s[0][cr & 63]
                else:
                    La, Lb = s[2 * k][cl & 63], s[2 * k + 1][cl & 63]
                    Ra, Rb = s[2 * k][cr & 63], s[2 * k + 1][cr & 63]
                    L = La if lane < 32 else Lb
                    R = Ra if lane < 31 else Rb
                    if 2 * k + 2 < ns and lane == 63:
                        R = s[2 * k + 2][0]
                hl[lane] = cl < end
                pr[lane] = cr < end and key(L) <= key(R)
                chosen[k][lane] = R if pr[lane] else L
            hasl[k], prefr[k] = ballot(hl), ballot(pr)
        pos = 0
        while True:
            k, b = pos >> 6, pos & 63
            if k >= kc:
                break
            if not (hasl[k] >> b) & 1:
                break
            pathm[k] |= 1 << b
            pos = 2 * pos + 1 + ((prefr[k] >> b) & 1)
        return chosen, pathm, pos

    def pop_with_last(self, last, end):
        root = self.get(0)
        chosen, pathm, bottom = self.chase(end)
        J = -1
        for k in range(self.ns):
            onpath = []
            for lane in LANES:
                i = 64 * k + lane
                op = i == bottom
                if k < self.kc:
                    op = op or (pathm[k] >> lane) & 1 != 0
                onpath.append(op and i != 0 and key(last) <= key(self.s[k][lane]))
            m = ballot(onpath)
            if m:
                J = 64 * k + highest(m)
        if J >= 0:
            new = [list(x) for x in self.s]
            for k in range(self.kc):
                for lane in LANES:
                    i = 64 * k + lane
                    if (pathm[k] >> lane) & 1 and i < J:
                        new[k][lane] = chosen[k][lane]
            self.s = new
        self.set(J if J >= 0 else 0, last)
        return root

    def pop(self, ln):
        last = self.get(ln - 1)
        ln -= 1
        if ln == 0:
            return last, ln
        return self.pop_with_last(last, ln), ln

    def push_then_pop_full(self, item):
        cap = self.cap
        cnt, j = 0, 1
        while ((cap + 1) >> j) >= 1:
            if key(item) <= key(self.get(((cap + 1) >> j) - 1)):
                break
            cnt = j
            j += 1
        last = item
        if cnt > 0:
            last = self.get(((cap + 1) >> 1) - 1)
            for j in range(1, cnt):
                self.set(((cap + 1) >> j) - 1, self.get(((cap + 1) >> (j + 1)) - 1))
            self.set(((cap + 1) >> cnt) - 1, item)
        self.pop_with_last(last, cap)

    def sift_down_range(self, end):
        elt = self.get(0)
        chosen, pathm, bottom = self.chase(end)
        F = -1
        for k in range(self.ns - 1, -1, -1):
            onpath = []
            for lane in LANES:
                i = 64 * k + lane
                op = i == bottom
                if k < self.kc:
                    op = op or (pathm[k] >> lane) & 1 != 0
                onpath.append(op and i != 0 and key(elt) >= key(self.s[k][lane]))
            m = ballot(onpath)
            if m:
                F = 64 * k + ctz(m)
        dest = ((F - 1) >> 1) if F >= 0 else bottom
        new = [list(x) for x in self.s]
        for k in range(self.kc):
            for lane in LANES:
                i = 64 * k + lane
                if (pathm[k] >> lane) & 1 and i < dest:
                    new[k][lane] = chosen[k][lane]
        self.s = new
        self.set(dest, elt)

    def array(self, ln):
        return [self.get(i) for i in range(ln)]


@pytest.mark.parametrize("ns", [1, 2, 4])
@pytest.mark.parametrize("ties", [False, True])
def test_register_heap_equals_std(ns, ties):
    """return_points as the search keeps it: pushes up to ef entries, then push + pop for every further one (ef == CAP: the fused
    form; ef < CAP: push then pop), pops in between, and at the end into_sorted_vec by sift_down_range -- against std."""
    rnd = random.Random(500 + 10 * ns + ties)
    k = _keys(rnd, ties)
    cap = 64 * ns
    for ef in sorted({1, 2, 3, 7, cap // 2, cap - 2, cap - 1, cap}):
        if ef < 1:
            continue
        for rep in range(6):
            a, h, ln, t = _StdBinaryHeap(), DevRegHeap(ns), 0, 0
            for _ in range(rnd.choice([ef, 3 * ef + 5, 600])):
                r = rnd.random()
                if ln and r < 0.15:
                    got, ln = h.pop(ln)
                    assert a.pop() == got
                else:
                    it = (k(), t)
                    t += 1
                    if ln == cap:                     # (only when ef == CAP)
                        a.push(it)
                        a.pop()
                        h.push_then_pop_full(it)
                    else:
                        a.push(it)
                        ln = h.push(ln, it)
                        if ln > ef:
                            got, ln = h.pop(ln)
                            assert a.pop() == got
                assert a.d == h.array(ln), (ns, ef, rep)
            # into_sorted_vec
            end = ln
            d = list(a.d)
            want = a.into_sorted_vec()
            while end > 1:
                end -= 1
                x, y = h.get(0), h.get(end)
                h.set(0, y)
                h.set(end, x)
                h.sift_down_range(end)
            assert h.array(ln) == want, (ns, ef, rep, d)
