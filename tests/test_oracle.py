"""CPU tests of the oracle itself (no GPU): what pins the restatement, since the reference's own tests
hold no golden vectors.  Known answers below are derived BY HAND from the published
algorithms (Rust std BinaryHeap; anndists scalar distances), not from running this code."""
import numpy as np
import pytest

from conftest import normalized, uniform


# ---------------------------------------------------------------- std::collections::BinaryHeap
def test_heap_pop_is_descending(oracle):
    rng = np.random.default_rng(0)
    v = rng.random(200).astype(np.float32)
    out, _ = oracle.heap_exercise(v, np.arange(200), mode=0, npop=200)
    assert np.array_equal(out, np.sort(v)[::-1])


def test_heap_into_sorted_vec_is_ascending(oracle):
    rng = np.random.default_rng(1)
    v = rng.random(137).astype(np.float32)
    out, tags = oracle.heap_exercise(v, np.arange(137), mode=1)
    assert np.array_equal(out, np.sort(v))
    assert np.array_equal(v[tags], out)


def test_heap_tie_order_hand_derived(oracle):
    """Three equal keys pushed as a, b, c.
    push: sift_up stops on `element <= parent`, so the array stays [a, b, c].
    pop #1: swap-remove root with last -> [c, b], popped a; sift_down_to_bottom moves b up -> [b, c].
    pop #2: -> b, pop #3: -> c.  Pop order a, b, c.
    into_sorted_vec on [a, b, c]: swap(0,2) -> [c,b,a], sift_down_range(0,2): child==end-1 and c<b false;
    swap(0,1) -> [b,c,a].  Sorted-vec order among equals: b, c, a."""
    vals = np.ones(3, np.float32)
    _, tags = oracle.heap_exercise(vals, [0, 1, 2], mode=0, npop=3)
    assert tags.tolist() == [0, 1, 2]
    _, tags = oracle.heap_exercise(vals, [0, 1, 2], mode=1)
    assert tags.tolist() == [1, 2, 0]


def test_heap_mixed_keys_hand_derived(oracle):
    """keys 2(a) 3(b) 3(c) 1(d): push a:[a]; push b: b>a -> [b,a]; push c: c<=b -> [b,a,c];
    push d: parent(3)=1 is a: d<=a -> [b,a,c,d].
    pop: last=d, swap root -> [d,a,c], popped b; sift_down_to_bottom(0): end=3, child=1: a<=c -> child=2,
    move c up -> [c,a,_], hole at 2; child=5 > end-2; child==end-1? 5!=2; sift_up(0,2): d<=c -> stays: [c,a,d].
    pop order: b, then c, then a, then d."""
    _, tags = oracle.heap_exercise(np.array([2, 3, 3, 1], np.float32), [0, 1, 2, 3], mode=0, npop=4)
    assert tags.tolist() == [1, 2, 0, 3]


# ---------------------------------------------------------------- anndists distances
def _seq32(vals):
    acc = np.float32(0)
    for v in vals:
        acc = np.float32(acc + np.float32(v))
    return acc


@pytest.mark.parametrize("d", [1, 2, 25, 128, 784])
def test_distances_are_left_to_right_f32_sums(oracle, d):
    rng = np.random.default_rng(d)
    a = (rng.random(d, dtype=np.float32) - 0.4) * 2
    b = (rng.random(d, dtype=np.float32) - 0.4) * 2
    t = (a - b).astype(np.float32)
    want_l2 = np.sqrt(_seq32((t * t).astype(np.float32)))
    assert np.float32(oracle.dist_eval("DistL2", a, b)).view(np.uint32) == np.float32(want_l2).view(np.uint32)
    want_l1 = _seq32(np.abs(t))
    assert np.float32(oracle.dist_eval("DistL1", a, b)).view(np.uint32) == np.float32(want_l1).view(np.uint32)
    an, bn = a / np.linalg.norm(a), b / np.linalg.norm(b)
    want_dot = max(np.float32(1) - _seq32((an * bn).astype(np.float32)), np.float32(0))
    assert np.float32(oracle.dist_eval("DistDot", an, bn)).view(np.uint32) == np.float32(want_dot).view(np.uint32)
    # cosine: f32 products, f64 sums
    s0 = sum(float(np.float32(x * y)) for x, y in zip(a, b))
    s1 = sum(float(np.float32(x * x)) for x in a)
    s2 = sum(float(np.float32(y * y)) for y in b)
    want_cos = np.float32(max(1.0 - s0 / np.sqrt(s1 * s2), 0.0))
    assert np.float32(oracle.dist_eval("DistCosine", a, b)).view(np.uint32) == want_cos.view(np.uint32)


def test_distance_known_values(oracle):
    assert oracle.dist_eval("DistL2", [3, 0], [0, 4]) == 5.0  # a true metric, not squared
    assert oracle.dist_eval("DistL1", [1, -2, 3], [0, 0, 0]) == 6.0
    assert oracle.dist_eval("DistCosine", [1, 0], [0, 1]) == 1.0
    assert oracle.dist_eval("DistCosine", [2, 0], [5, 0]) == 0.0
    assert oracle.dist_eval("DistCosine", [0, 0], [1, 1]) == 0.0  # zero norm => 0
    assert oracle.dist_eval("DistDot", [1, 0], [1, 0]) == 0.0
    # tests/filtertest.rs:248 uses DistL2 < 1e-2 as a distance threshold on a unit grid
    assert oracle.dist_eval("DistL2", [1.0, 1.0], [1.0, 1.005]) < 1e-2


# ---------------------------------------------------------------- search properties (SURVEY.md section 4 list)
@pytest.fixture(scope="module")
def small_index(oracle):
    X = uniform(3000, 16, 42)
    o = oracle.OracleHnsw(16, 3000, 16, 200, "DistL2")
    o.insert_batch(X)
    return X, o


def test_results_sorted_and_bounded(small_index):
    X, o = small_index
    Q = uniform(100, 16, 43)
    r = o.parallel_search(Q, 10, 40, 4)
    assert np.all(r.counts == 10)
    assert np.all(np.diff(r.dists, axis=1) >= 0)


def test_self_query_distance_zero(small_index):
    X, o = small_index
    r = o.parallel_search(X[:300], 1, 64, 4)
    found = r.ids[:, 0] == np.arange(300)
    assert found.mean() > 0.95
    assert np.all(r.dists[found, 0] == 0.0)


def test_recall_against_brute_force(small_index):
    X, o = small_index
    Q = uniform(200, 16, 44)
    r = o.parallel_search(Q, 10, 64, 4)
    D = np.sqrt(((Q[:, None, :].astype(np.float64) - X[None]) ** 2).sum(-1))
    gt = np.argsort(D, axis=1)[:, :10]
    by_id = np.mean([len(set(gt[i]) & set(r.ids[i].tolist())) / 10 for i in range(200)])
    kth = np.sort(D, axis=1)[:, 9]
    by_dist = np.mean(r.dists <= kth[:, None] * (1 + 1e-6))  # examples/ann-sift1m-128-euclidean.rs:172-186
    assert by_id > 0.9 and by_dist > 0.9


def test_serial_equals_parallel(small_index):
    X, o = small_index
    Q = uniform(50, 16, 45)
    r = o.parallel_search(Q, 7, 30, 8)
    for i in range(50):
        ids, dists, _, _ = o.search(Q[i], 7, 30)
        assert np.array_equal(ids, r.ids[i]) and np.array_equal(dists, r.dists[i])


def test_k_larger_than_ef(small_index):
    X, o = small_index
    r = o.parallel_search(uniform(10, 16, 46), 50, 5, 2)  # ef = max(ef, knbn) (src/hnsw.rs:1531)
    assert np.all(r.counts == 50)


def test_sparse_search(oracle):
    """src/hnsw.rs:1870-1881: one 4-d point, search(k=2, ef=10) returns exactly it at distance 0,
    whichever layer the level generator drew."""
    seen_layers = set()
    for i in range(60):
        o = oracle.OracleHnsw(3, 10, 16, 20, "DistL2")  # small M => frequent upper layers
        X = uniform(1 + i, 4, i)  # insert i extra points so the LAST one lands in varying layers
        o.insert_batch(X[:1])
        ids, dists, layers, _ = o.search(X[0], 2, 10)
        assert len(ids) == 1 and dists[0] == 0.0 and ids[0] == 0
        seen_layers.add(int(layers[0]))
    assert 0 in seen_layers


def test_empty_index_returns_nothing(oracle):
    o = oracle.OracleHnsw(8, 10, 16, 20, "DistL2")
    r = o.parallel_search(uniform(3, 4, 1), 5, 10, 1)
    assert np.all(r.counts == 0)


def test_dump_reload_dump_and_searches_alike(small_index, oracle, tmp_path):
    import filecmp
    X, o = small_index
    o.file_dump(tmp_path, "a")
    o2 = oracle.OracleHnsw.load(tmp_path, "a", "DistL2")
    o2.file_dump(tmp_path, "b")
    from conftest import same_dump_after_reload
    assert same_dump_after_reload(tmp_path / "a.hnsw.graph", tmp_path / "b.hnsw.graph")  # all bytes but the level scale
    assert filecmp.cmp(tmp_path / "a.hnsw.data", tmp_path / "b.hnsw.data", shallow=False)
    Q = uniform(40, 16, 47)
    r1, r2 = o.parallel_search(Q, 10, 32, 2), o2.parallel_search(Q, 10, 32, 2)
    assert np.array_equal(r1.ids, r2.ids) and np.array_equal(r1.dists, r2.dists)
    with pytest.raises(RuntimeError):
        oracle.OracleHnsw.load(tmp_path, "a", "DistCosine")  # distance short-name check (src/hnswio.rs:473-490)


def test_level_law(oracle):
    """P(level >= l) = M^-l for the default scale (src/hnsw.rs:356-360)."""
    lv = oracle.levels(16, 200000)
    frac1 = (lv >= 1).mean()
    assert abs(frac1 - 1 / 16) < 0.004
    assert abs((lv >= 2).mean() - 1 / 256) < 0.001
    lv2 = oracle.levels(16, 200000, scale_factor=0.5)  # modify_level_scale(0.5): P(l>=1) = M^-2
    assert abs((lv2 >= 1).mean() - 1 / 256) < 0.001


# ---- the distances between probability vectors (f32 arms of src/libext.rs:334-345, :491-513) and f32::ln
def test_ref_logf_is_the_hosts_logf(oracle):
    """f32::ln is the platform libm's logf; oracle/ref_logf.hpp restates glibc's (table + degree-3 polynomial in double).
    Exhaustively (all 2 139 095 039 positive finite floats, 8 threads, 6 s) the restatement equals glibc 2.35's logf bit for
    bit, with and without FMA contraction; the suite re-checks a stride of it, every float within 2^20 ulps of 1.0 (where
    the cancellation is), the subnormals' edge and the special values."""
    L = oracle.lib()
    assert L.orc_ref_logf_mismatches(0x00000001, 0x7F7FFFFF, 211) == 0
    assert L.orc_ref_logf_mismatches(0x3F800000 - (1 << 20), 0x3F800000 + (1 << 20), 1) == 0
    assert L.orc_ref_logf_mismatches(0x00000001, 0x00900000, 13) == 0
    assert L.orc_ref_logf(1.0) == 0.0 and L.orc_ref_logf(0.0) == -np.inf and np.isnan(L.orc_ref_logf(-1.0))
    assert L.orc_ref_logf(np.inf) == np.inf


def _prob(n, d, seed):
    x = np.random.default_rng(seed).random((n, d), dtype=np.float32) + np.float32(1e-3)
    x[:, ::7] = 0.0                                       # exact zeros are legal in a distribution
    return (x / x.sum(1, dtype=np.float32)[:, None]).astype(np.float32)


def test_probability_distances_follow_the_formulas_step_by_step(oracle):
    """anndists 0.1 (recalled, oracle/PIN.md): Hellinger sqrt(max(1 - sum sqrt(a) sqrt(b), 0)); Jeffreys
    sum (a-b) ln(max(a,1e-30)/max(b,1e-30)); JensenShannon sqrt(0.5 sum [a ln(a/m) if a>0] + [b ln(b/m) if b>0]) -- every
    operation a separate f32 rounding, sums left to right."""
    f = np.float32
    ln = lambda v: f(oracle.lib().orc_ref_logf(float(v)))
    P = _prob(6, 19, 3)
    for i in range(5):
        a, b = P[i], P[i + 1]
        s = f(0)
        for x, y in zip(a, b):
            s = f(s + f(np.sqrt(x, dtype=f) * np.sqrt(y, dtype=f)))
        assert oracle.dist_eval("DistHellinger", a, b) == f(np.sqrt(max(f(f(1) - s), f(0)), dtype=f))
        s = f(0)
        for x, y in zip(a, b):
            s = f(s + f(f(x - y) * ln(f(max(x, f(1e-30)) / max(y, f(1e-30))))))
        assert oracle.dist_eval("DistJeffreys", a, b) == s
        s = f(0)
        for x, y in zip(a, b):
            m = f(f(0.5) * f(x + y))
            if x > 0:
                s = f(s + f(x * ln(f(x / m))))
            if y > 0:
                s = f(s + f(y * ln(f(y / m))))
        assert oracle.dist_eval("DistJensenShannon", a, b) == f(np.sqrt(f(f(0.5) * s), dtype=f))
    for dist in ("DistHellinger", "DistJeffreys", "DistJensenShannon"):
        assert abs(oracle.dist_eval(dist, P[0], P[0])) < 4e-4     # d(p, p) = 0 (Hellinger: up to the rounding of sum sqrt(p)^2)


class _StdBinaryHeap:
    """A second, independent transcription of Rust std's `BinaryHeap` (alloc/collections/binary_heap.rs), written from the
    algorithm's published description and kept apart from oracle/hnsw_oracle.hpp on purpose: `push` = append + `sift_up`
    (stop at `elt <= parent`), `pop` = swap-remove the root with the last element, `sift_down_to_bottom` (always down to a
    leaf, taking the right child when `left <= right`), then `sift_up`; `into_sorted_vec` = repeated swap of root and end
    with `sift_down_range` (stop at `elt >= child`; a lone last child is taken when `elt < child`).  Entries are
    (key, tag); only the key is compared."""

    def __init__(self):
        self.d = []

    def _sift_up(self, start, pos):
        elt = self.d[pos]
        while pos > start:
            parent = (pos - 1) // 2
            if elt[0] <= self.d[parent][0]:
                break
            self.d[pos] = self.d[parent]
            pos = parent
        self.d[pos] = elt
        return pos

    def push(self, item):
        self.d.append(item)
        self._sift_up(0, len(self.d) - 1)

    def pop(self):
        if not self.d:
            return None
        item = self.d.pop()
        if self.d:
            item, self.d[0] = self.d[0], item
            end, pos = len(self.d), 0
            elt = self.d[0]
            child = 1
            while child <= max(end, 2) - 2 and end >= 2:
                if self.d[child][0] <= self.d[child + 1][0]:
                    child += 1
                self.d[pos] = self.d[child]
                pos = child
                child = 2 * pos + 1
            if child == end - 1:
                self.d[pos] = self.d[child]
                pos = child
            self.d[pos] = elt
            self._sift_up(0, pos)
        return item

    def into_sorted_vec(self):
        d = self.d
        end = len(d)
        while end > 1:
            end -= 1
            d[0], d[end] = d[end], d[0]
            pos, elt = 0, d[0]
            child = 1
            moved = False
            while child <= max(end, 2) - 2 and end >= 2:
                if d[child][0] <= d[child + 1][0]:
                    child += 1
                if elt[0] >= d[child][0]:
                    d[pos] = elt
                    moved = True
                    break
                d[pos] = d[child]
                pos = child
                child = 2 * pos + 1
            if not moved:
                if child == end - 1 and elt[0] < d[child][0]:
                    d[pos] = d[child]
                    pos = child
                d[pos] = elt
        return d


@pytest.mark.parametrize("seed", range(12))
def test_two_transcriptions_of_std_binary_heap_agree(oracle, seed):
    """Random scripts of pushes and pops over FEW distinct keys (ties everywhere, tags tell equal keys apart): the oracle's C++
    `RustBinaryHeap` and the Python transcription above must pop the same tags in the same order and leave the same
    `into_sorted_vec`.  Not a pin (both are restatements), but a transcription slip in either shows up here."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(50, 900))
    levels = int(rng.choice([2, 3, 5, 17, 1000]))
    vals = (rng.integers(0, levels, n) / np.float32(levels)).astype(np.float32)
    tags = np.arange(n, dtype=np.int32)
    is_pop = (rng.random(n) < rng.choice([0.15, 0.35, 0.5])).astype(np.uint8)
    pv, pt, sv, st = oracle.heap_script(vals, tags, is_pop)
    h = _StdBinaryHeap()
    popped = []
    for i in range(n):
        if is_pop[i]:
            x = h.pop()
            if x is not None:
                popped.append(x)
        else:
            h.push((float(vals[i]), int(tags[i])))
    rest = h.into_sorted_vec()
    assert [t for _, t in popped] == list(pt) and np.array_equal(np.array([v for v, _ in popped], np.float32), pv)
    assert [t for _, t in rest] == list(st) and np.array_equal(np.array([v for v, _ in rest], np.float32), sv)
