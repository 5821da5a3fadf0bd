"""heap_push_batch16 (hnswlib-rs_amd/csrc/search_kernels.inc): several BinaryHeap::push in a row with one round trip to the heap's
memory -- every node of the union of the new leaves' root paths gets a lane (16 lanes for the leaves, then 9, 5, 3 and 2 per
level), the sift-ups run one after the other on those registers, what changed is stored back.  The lane algorithm is emulated
here line by line and fuzzed against std's push (the Python transcription of tests/test_oracle.py), with pops in between and
tie-heavy keys; the rule that keeps a run on ONE level of the tree (no power of two among the new positions) is what the
round-4 soak found missing: without it the root sits in two lanes."""
import random

import pytest

from test_oracle import _StdBinaryHeap


def _lane0(j):
    return 0 if j == 0 else 16 if j == 1 else 25 if j == 2 else 30 if j == 3 else 33 + 2 * (j - 4)


def _batch16(data, items):
    """the device routine on a Python list `data` (heap array of (key, tag)); items in push order"""
    m, ln = len(items), len(data)
    p0, pl = ln + 1, ln + m
    node, valid, v = [0] * 64, [False] * 64, [None] * 64
    for l in range(64):
        j = 0 if l < 16 else 1 if l < 25 else 2 if l < 30 else 3 if l < 33 else 4 + ((l - 33) >> 1)
        node[l] = (p0 >> j) + (l - _lane0(j))
        valid[l] = 1 <= node[l] <= (pl >> j)
        v[l] = data[node[l] - 1] if valid[l] and j >= 1 else (0.0, -1)
    touched = set()
    for i, item in enumerate(items):
        cur, jc, lane_cur = p0 + i, 0, i
        while True:
            par = cur >> 1
            if par == 0:
                break
            lane_par = _lane0(jc + 1) + (par - (p0 >> (jc + 1)))
            assert valid[lane_par] and node[lane_par] == par
            if item[0] <= v[lane_par][0]:      # sift_up stops on <=
                break
            v[lane_cur] = v[lane_par]
            touched.add(lane_cur)
            cur, jc, lane_cur = par, jc + 1, lane_par
        v[lane_cur] = item
        touched.add(lane_cur)
    data.extend([None] * m)
    for l in touched:
        assert valid[l]
        data[node[l] - 1] = v[l]


def _push_run(data, items, one_level_rule=True):
    """heap_push_batch's dispatch: the batch routine when it applies, else one push at a time"""
    p0, pl = len(data) + 1, len(data) + len(items)
    ok = len(data) >= 64 and len(data) + 16 < (1 << 19) and len(items) >= 2
    if one_level_rule:
        ok = ok and p0.bit_length() == pl.bit_length()
    if ok:
        _batch16(data, items)
    else:
        h = _StdBinaryHeap()
        h.d = data
        for it in items:
            h.push(it)


@pytest.mark.parametrize("ties", [False, True])
def test_batched_pushes_equal_std_pushes(ties):
    rnd = random.Random(11 + ties)
    batches = 0
    for rep in range(40):
        a, b = _StdBinaryHeap(), _StdBinaryHeap()
        t = 0
        for _ in range(rnd.randint(100, 700)):
            m = rnd.randint(1, 16)
            items = [((rnd.choice([0, 1, 2, 3, 5, 8, 9]) if ties else rnd.random()), t + i) for i in range(m)]
            t += m
            for it in items:
                a.push(it)
            _push_run(b.d, items)
            batches += 1
            assert a.d == b.d
            for _ in range(rnd.randint(0, 3)):
                if a.d:
                    assert a.pop() == b.pop()
    assert batches > 10_000


def test_a_run_across_a_power_of_two_needs_the_one_level_rule():
    """506 entries + 10 pushes: the new positions 507..516 cross 512, the root is the 8th ancestor of 507..511 and the 9th of
    512..516 -- without the rule some item that climbs to the root is lost from one of its two lanes."""
    rnd = random.Random(5)
    broke = False
    for _ in range(400):
        a, b = _StdBinaryHeap(), _StdBinaryHeap()
        for t in range(506):
            x = (rnd.random(), t)
            a.push(x)
            b.push(x)
        items = [(rnd.random() * 1.2, 1000 + i) for i in range(10)]
        for it in items:
            a.push(it)
        try:
            _push_run(b.d, items, one_level_rule=False)
            broke = broke or a.d != b.d
        except AssertionError:
            broke = True
    assert broke
