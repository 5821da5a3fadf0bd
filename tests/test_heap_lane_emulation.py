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
