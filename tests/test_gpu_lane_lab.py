"""The wave-level algorithms of the search kernels, run ON THE DEVICE from scripts (hnswgpu_lane_lab, csrc/lane_lab.inc) and
fuzzed against referees that are not transcriptions of the lane code:

* the memory heap (heap_push / heap_push_batch -> heap_push_batch16 / heap_pop / heap_pop3 / heap_sift_down_range) and the register
  heap (RegHeap push / pop / push_then_pop_full / sift_down_range) against the oracle's RustBinaryHeap (std's push / pop /
  into_sorted_vec, src/hnsw.rs:283-297 for the order) -- interleaved pushes and pops, keys full of ties, runs of batched pushes that
  cross powers of two (the round-4 bug), heaps that straddle their LDS part;
* the result set (r_insert, merge_list = csrc/merge_list_body.inc, the very text the search kernel includes) against the accept
  rule taken one neighbour at a time (src/hnsw.rs:1028-1053) on a Python list;
* the 16-bit-cell visited table (cell16_test / cell16_insert / visit_cell16, the kernel's test-all-then-insert protocol) against a
  Python set, the table decoded back to ids at the end.

The Python models of rounds 3-4 (tests/test_heap_lane_emulation.py, test_result_set_emulation.py, test_visited_table_emulation.py,
test_heap_batch_push.py) stay as what they are -- models, CPU tests -- and lend this file their decoders and generators."""
import random

import numpy as np
import pytest

from test_visited_table_emulation import Table

pytestmark = pytest.mark.gpu

PUSH, POP, PUSH_LANES, INSERT, MERGE, BATCH, VISIT = 1, 2, 3, 4, 5, 6, 7


def f2u(x):
    return int(np.float32(x).view(np.uint32))


def u2f(w):
    return float(np.uint32(w).view(np.float32))


def run_lab(native, mode, p0, p1, p2, ops, lanes=None, out_words=1 << 16):
    lib = native.lib()
    ops = np.ascontiguousarray(np.array(ops, dtype=np.uint32).reshape(-1, 4))
    lanes = np.zeros((0, 64, 2), np.uint32) if lanes is None else np.ascontiguousarray(np.array(lanes, dtype=np.uint32).reshape(-1, 64, 2))
    out = np.zeros(out_words, np.uint32)
    rc = lib.hnswgpu_lane_lab(0, mode, p0, p1, p2, ops.ctypes.data, len(ops), lanes.ctypes.data, len(lanes), out.ctypes.data, out_words)
    assert rc == 0, native._native.last_error()
    assert 1 <= out[0] <= out_words, f"the script produced {out[0]} words, the buffer holds {out_words}"
    return out[1:out[0]].tolist()


def keys(rnd, n, ties):
    """f32 keys: many exact ties (small integer grid) or few"""
    if ties:
        return [np.float32(rnd.randrange(0, max(2, n // 6))) * np.float32(0.25) for _ in range(n)]
    return [np.float32(rnd.random()) for _ in range(n)]


# ------------------------------------------------------------------------------------------------- BinaryHeap in memory
def heap_case(rnd, n_ops, ties, batches, cap=None):
    """A script of pushes (single and, `batches`, runs of <= 64 through heap_push_batch), and pops that never meet an empty heap.
    Returns (device ops, lane sets, the same script for the oracle: vals, tags, is_pop)."""
    ops, lanes, vals, tags, is_pop = [], [], [], [], []
    size, tag = 0, 0
    ks = keys(rnd, 4 * n_ops + 256, ties)
    ki = 0
    while len(ops) < n_ops:
        r = rnd.random()
        if size > 0 and r < 0.3:
            ops.append((POP, 0, 0, 0))
            vals.append(0.0); tags.append(0); is_pop.append(1)
            size -= 1
        elif batches and r < 0.55:
            m = rnd.choice([1, 2, 3, 5, 16, 17, 31, 40, 64])
            if cap is not None and size + m > cap:
                continue
            lanes_used = sorted(rnd.sample(range(64), m))
            mask = sum(1 << l for l in lanes_used)
            vec = np.zeros((64, 2), np.uint32)
            for l in range(64):
                vec[l, 0] = f2u(-ks[ki + l])  # the batch pushes {-de, idc}: de = -key
                vec[l, 1] = 0x7000000 + l
            for l in lanes_used:
                vec[l, 1] = tag
                vals.append(float(ks[ki + l])); tags.append(tag); is_pop.append(0)
                tag += 1
            ki += 64
            if ki + 64 > len(ks):
                ks += keys(rnd, 1024, ties)
            ops.append((PUSH_LANES, mask & 0xFFFFFFFF, mask >> 32, len(lanes)))
            lanes.append(vec)
            size += m
        else:
            if cap is not None and size + 1 > cap:
                continue
            k = ks[ki]; ki += 1
            ops.append((PUSH, f2u(k), tag, 0))
            vals.append(float(k)); tags.append(tag); is_pop.append(0)
            tag += 1
            size += 1
    return ops, lanes, vals, tags, is_pop


def check_heap(out, oracle, vals, tags, is_pop, silent_pops=()):
    pv, pt, sv, st = oracle.heap_script(vals, tags, is_pop)
    keep = [i for i in range(len(pv)) if i not in silent_pops]
    n_pop = len(keep)
    got_pops = [(u2f(out[2 * i]), out[2 * i + 1]) for i in range(n_pop)]
    assert got_pops == [(float(pv[i]), int(pt[i])) for i in keep], "popped entries differ from std's BinaryHeap"
    rest = out[2 * n_pop:]
    assert rest[0] == len(sv)
    got_sorted = [(u2f(rest[1 + 2 * i]), rest[2 + 2 * i]) for i in range(rest[0])]
    assert got_sorted == [(float(v), int(t)) for v, t in zip(sv, st)], "into_sorted_vec differs from std's BinaryHeap"


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("lds_cap,pop3", [(0, 0), (8, 1), (64, 1), (512, 0), (512, 1), (1024, 1)])
def test_memory_heap_equals_std_binaryheap(native, oracle, lds_cap, pop3, ties):
    """heap_push / heap_pop / heap_pop3 / heap_sift_down_range on the device == Rust's BinaryHeap, heaps that live in LDS, in the global
    slice, and across the border between the two."""
    rnd = random.Random(1000 * lds_cap + 10 * pop3 + int(ties))
    for n_ops in (5, 40, 300, 1500):
        ops, lanes, vals, tags, is_pop = heap_case(rnd, n_ops, ties, batches=False)
        check_heap(run_lab(native, 0, lds_cap, pop3, 0, ops), oracle, vals, tags, is_pop)
        # the same script with the pushes a log replay uses (no parent probe in front of the ancestor chain)
        check_heap(run_lab(native, 0, lds_cap, pop3, 1, ops), oracle, vals, tags, is_pop)


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("lds_cap", [0, 32, 512])
def test_batched_pushes_equal_std_binaryheap(native, oracle, lds_cap, ties):
    """heap_push_batch (runs of <= 16 through heap_push_batch16, one level of the tree per run, single pushes across a power of two --
    the case the round-4 soak caught) == the same entries pushed one by one into Rust's BinaryHeap, interleaved with pops."""
    rnd = random.Random(77 + lds_cap + int(ties))
    for n_ops in (8, 60, 400):
        ops, lanes, vals, tags, is_pop = heap_case(rnd, n_ops, ties, batches=True)
        check_heap(run_lab(native, 0, lds_cap, 1, 0, ops, lanes), oracle, vals, tags, is_pop)
    # runs that end exactly on, start exactly at, and straddle powers of two
    for start in (14, 15, 16, 30, 31, 32, 62, 63, 64, 127, 250, 255, 256, 511, 1020):
        ops, lanes, vals, tags, is_pop = [], [], [], [], []
        ks = keys(rnd, start + 64, ties)
        for t in range(start):
            ops.append((PUSH, f2u(ks[t]), t, 0)); vals.append(float(ks[t])); tags.append(t); is_pop.append(0)
        m = rnd.choice([2, 5, 16, 33])
        vec = np.zeros((64, 2), np.uint32)
        for l in range(64):
            vec[l] = (f2u(-ks[start + l]), 0x7000000 + l)
        for l in range(m):
            vec[l, 1] = start + l
            vals.append(float(ks[start + l])); tags.append(start + l); is_pop.append(0)
        ops.append((PUSH_LANES, (1 << m) - 1 & 0xFFFFFFFF, ((1 << m) - 1) >> 32, 0))
        lanes.append(vec)
        for _ in range(3):
            ops.append((POP, 0, 0, 0)); vals.append(0.0); tags.append(0); is_pop.append(1)
        check_heap(run_lab(native, 0, lds_cap, 1, 0, ops, lanes), oracle, vals, tags, is_pop)


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("ns", [1, 2, 4])
def test_register_heap_equals_std_binaryheap(native, oracle, ns, ties):
    """RegHeap<NS> (return_points of a literal search: push, pop, the fused push-then-pop of a full heap, sift_down_range) == Rust's
    BinaryHeap; a push into a full heap (64 NS entries) is `push, then pop` (src/hnsw.rs:1038, :1051-1053), its pop not reported."""
    rnd = random.Random(31 * ns + int(ties))
    cap = 64 * ns
    for n_ops in (6, 50, 3 * cap, 8 * cap):
        ops, vals, tags, is_pop, silent = [], [], [], [], set()
        size, tag, n_pops = 0, 0, 0
        ks = keys(rnd, n_ops + 8, ties)
        for t in range(n_ops):
            if size > 0 and rnd.random() < (0.15 if size < cap else 0.05):
                ops.append((POP, 0, 0, 0)); vals.append(0.0); tags.append(0); is_pop.append(1)
                size -= 1
                n_pops += 1
            else:
                ops.append((PUSH, f2u(ks[t]), tag, 0)); vals.append(float(ks[t])); tags.append(tag); is_pop.append(0)
                tag += 1
                if size == cap:  # fused: the pop that follows is part of the push
                    vals.append(0.0); tags.append(0); is_pop.append(1)
                    silent.add(n_pops)
                    n_pops += 1
                else:
                    size += 1
        check_heap(run_lab(native, 1, ns, 0, 0, ops), oracle, vals, tags, is_pop, silent)


# ------------------------------------------------------------------------------------------------- the result set
def seq_accept(R, ef, xd, xi):
    """src/hnsw.rs:1028-1053 on the sorted-array form: accepted iff the set is not full or xd < the farthest entry; the entry goes
    behind everything not farther (arrival order among equals); the entry pushed past ef - 1 leaves.  Returns (accepted, tie)."""
    if len(R) >= ef and not xd < R[-1][0]:
        return False, False
    tie = any(d == xd for d, _ in R)
    pos = sum(1 for d, _ in R if d <= xd)
    R.insert(pos, (xd, xi))
    del R[ef:]
    return True, tie


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("S,ef", [(1, 1), (1, 10), (1, 63), (1, 64), (2, 65), (2, 100), (2, 128), (4, 129), (4, 200), (4, 256)])
def test_result_set_insertions_and_whole_list_accepts(native, S, ef, ties):
    """r_insert one neighbour at a time, then lists of up to 64 neighbours through the kernel's accept step (merge_list when the set
    is full -- csrc/merge_list_body.inc, the text the search kernel includes -- one at a time otherwise or when merge_list declines
    because equal entries would straddle the cut): the array, the tie flags and the accepted lanes == the sequential rule."""
    rnd = random.Random(100 * S + ef + int(ties))
    n_merged = 0
    for round_ in range(6):
        ops, lanes, R, expect = [], [], [], []
        n_single = rnd.choice([0, 3, ef // 2, ef + 5])
        nid = 1
        kgen = (lambda: np.float32(rnd.randrange(0, 40)) * np.float32(0.5)) if ties else (lambda: np.float32(rnd.random() * 20))
        for _ in range(n_single):
            xd = kgen()
            # r_insert itself is unconditional (the caller filters): a non-qualifying entry falls off the end again
            tie = any(d == float(xd) for d, _ in R)
            pos = sum(1 for d, _ in R if d <= float(xd))
            R.insert(pos, (float(xd), nid))
            del R[ef:]
            ops.append((INSERT, f2u(xd), nid, 0))
            expect.append(("ins", int(tie)))
            nid += 1
        for _ in range(rnd.choice([3, 12, 30])):
            nv = rnd.choice([1, 2, 7, 16, 32, 48, 64])
            vec = np.zeros((64, 2), np.uint32)
            lanes_used = sorted(rnd.sample(range(64), nv))
            # late in a search most neighbours are farther than the farthest entry: mix near and far keys
            far = R[-1][0] if R else 10.0
            acc_lanes, tie_any = [], False
            was_full = len(R) >= ef
            for l in range(64):
                xd = kgen() if rnd.random() < 0.5 else np.float32(far + rnd.randrange(0, 3) * (0.5 if ties else rnd.random()))
                vec[l] = (f2u(xd), 0x7000000 + l)
            for l in lanes_used:
                vec[l, 1] = nid
                nid += 1
            for l in lanes_used:
                a, t = seq_accept(R, ef, u2f(vec[l, 0]), int(vec[l, 1]))
                if a:
                    acc_lanes.append(l)
                    tie_any = tie_any or t
            mask = sum(1 << l for l in lanes_used)
            ops.append((MERGE, mask & 0xFFFFFFFF, mask >> 32, len(lanes)))
            lanes.append(vec)
            expect.append(("merge", was_full, sum(1 << l for l in acc_lanes), tie_any))
        out = run_lab(native, 2, S, ef, 0, ops, lanes)
        p = 0
        for e in expect:
            if e[0] == "ins":
                assert out[p] == e[1], "r_insert's tie flag"
                p += 1
            else:
                taken, tie, lo, hi = out[p:p + 4]
                p += 4
                _, was_full, acc, tie_any = e
                assert taken in ((0, 1) if was_full and acc else (0, 1, 2))
                if taken == 1:  # merge_list took the list: the lanes it accepted and its tie flag are the sequential rule's
                    n_merged += 1
                    assert (hi << 32) | lo == acc, "lanes accepted by merge_list"
                    if acc:
                        assert tie == int(tie_any), "merge_list's tie flag"
        assert out[p] == len(R)
        got = [(u2f(out[p + 1 + 2 * i]), out[p + 2 + 2 * i]) for i in range(out[p])]
        assert got == R, "the result set after the script"
    assert n_merged > 0, "merge_list never took a list: the scripts do not test it"


# ------------------------------------------------------------------------------------------------- the visited table
@pytest.mark.parametrize("tbits,idbits", [(6, 10), (8, 16), (9, 12), (11, 20), (11, 21), (12, 20), (12, 22)])
def test_visited_table_is_exact_under_the_batch_protocol(native, tbits, idbits):
    """cell16_test on every id of a batch, visit_cell16 for the ids whose home bucket is full, cell16_insert of the fresh ones from the
    (by then stale) snapshot of their test -- lanes racing for the same words -- == a Python set: `fresh` exactly for the ids not seen
    before, `no room` never below the kernel's load limit, and the table decoded at the end holds exactly the ids inserted."""
    rnd = random.Random(tbits * 100 + idbits)
    cells = 1 << tbits
    ops, lanes, batches = [], [], []
    universe = 1 << idbits
    first = rnd.randrange(universe)
    ops.append((VISIT, first, 0, 0))
    ops.append((VISIT, first, 0, 0))
    limit = cells - cells // 4  # the kernel stops inserting at 75 % load (the query then moves to the HBM bitmap)
    hot = [rnd.randrange(universe) for _ in range(200)]
    offered = {first}
    while len(offered) + 64 <= limit:
        nv = rnd.choice([1, 5, 16, 32, 48, 64])
        ids = set()
        while len(ids) < nv:  # a neighbour list holds no id twice; a third of the ids have been offered before
            r = rnd.random()
            ids.add(rnd.choice(hot) if r < 0.2 else (rnd.choice(tuple(offered)) if r < 0.35 else rnd.randrange(universe)))
        ids = list(ids)
        vec = np.zeros((64, 2), np.uint32)
        vec[:, 1] = 0xFFFFFFFF
        for l, i in enumerate(ids):
            vec[l, 1] = i
        offered.update(ids)
        ops.append((BATCH, nv, 0, len(lanes)))
        lanes.append(vec)
        batches.append(ids)
    out = run_lab(native, 3, tbits, idbits, idbits - (tbits - 3), ops, lanes, out_words=(1 << 13) + 8 * len(ops))
    assert out[0] == 1 and out[1] == 0, "visit_cell16: newly marked, then already visited"
    p = 2
    seen, n_noroom = {first}, 0
    for ids in batches:
        fm = out[p] | (out[p + 1] << 32)
        nm = out[p + 2] | (out[p + 3] << 32)
        p += 4
        unseen = sum(1 << l for l, i in enumerate(ids) if i not in seen)
        # a lane that found no room within its displacement budget (the kernel then moves the query's set to the HBM bitmap) has
        # not stored its id; everywhere else `fresh` is exactly "not seen before"
        assert nm & ~unseen == 0, "no room reported for an id that is in the table"
        assert fm & ~nm == unseen & ~nm, "fresh mask of a batch"
        n_noroom += bin(nm).count("1")
        seen.update(i for l, i in enumerate(ids) if not (nm >> l) & 1)
    assert n_noroom * 100 <= len(seen), "no room below the load limit is an exception"
    words = out[p]
    assert words == cells // 2
    t = Table(tbits, idbits)
    t.words = out[p + 1:p + 1 + words]
    stored = t.ids_stored()
    assert len(stored) == len(set(stored)) == len(seen) and set(stored) == seen
