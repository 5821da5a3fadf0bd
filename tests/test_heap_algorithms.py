"""The device emulates Rust's BinaryHeap with lane-parallel algorithms (hnswlib-rs_amd/csrc/search_kernels.inc):

  RegHeap<NS>       return_points in VGPRs, entry i in slot i // 64 of lane i % 64.  pop / sift_down_range: every
                    node compares its two children at once, the greater-child path is a walk over two bit masks,
                    the sift-up stop a third mask, the data movement one predicated move.
  heap_pop3         candidate_points in LDS/HBM: rounds of a 63-node subtree held by the lanes, five levels walked
                    per round, path nodes store their entry at their parent's index.
  heap_push_cached  push whose parent comparison uses a block of parents prefetched before the batch.

This file restates those algorithms lane by lane in Python (same index arithmetic, same masks) and checks them
against the oracle's literal BinaryHeap (oracle/hnsw_oracle.hpp, RustBinaryHeap) on interleaved pushes and pops
full of equal keys, where only the sift order decides.  It documents the algorithms and pins their logic on CPU;
the HIP code itself is covered by the -m gpu parity tests.
"""
import numpy as np
import pytest

MASK64 = (1 << 64) - 1


class RegHeapEmu:
    """RegHeap<NS>: s[k][lane] = entry 64 k + lane; entries are (key, tag), compared by key only."""

    def __init__(self, ns):
        self.ns = ns
        self.kc = 1 if ns == 1 else ns // 2   # slots whose nodes can have children below CAP
        self.cap = 64 * ns
        self.s = [[(0.0, -1)] * 64 for _ in range(ns)]
        self.len = 0

    def get(self, i):
        return self.s[i >> 6][i & 63]

    def set(self, i, v):
        self.s[i >> 6][i & 63] = v

    def push(self, item):            # serial sift_up (the device's RegHeap::push)
        pos = self.len
        self.len += 1
        while pos > 0:
            par = (pos - 1) >> 1
            pe = self.get(par)
            if item[0] <= pe[0]:
                break
            self.set(pos, pe)
            pos = par
        self.set(pos, item)

    def chase(self, end):
        chosen = [[None] * 64 for _ in range(self.kc)]
        hasl, prefr, pathm = [0] * self.kc, [0] * self.kc, [0] * self.kc
        for k in range(self.kc):
            for lane in range(64):
                i = 64 * k + lane
                cl, cr = 2 * i + 1, 2 * i + 2
                if self.ns == 1:
                    L, R = self.s[0][cl & 63], self.s[0][cr & 63]
                else:   # children of slot k live in slots 2k (lower lanes), 2k + 1, and slot 2k + 2 for lane 63's right child
                    L = self.s[2 * k][cl & 63] if lane < 32 else self.s[2 * k + 1][cl & 63]
                    R = self.s[2 * k][cr & 63] if lane < 31 else self.s[2 * k + 1][cr & 63]
                    if 2 * k + 2 < self.ns and lane == 63:
                        R = self.s[2 * k + 2][0]
                pr = cr < end and L[0] <= R[0]
                chosen[k][lane] = R if pr else L
                if cl < end:
                    hasl[k] |= 1 << lane
                if pr:
                    prefr[k] |= 1 << lane
        pos = 0
        while True:
            k, b = pos >> 6, pos & 63
            if k >= self.kc or not (hasl[k] >> b) & 1:
                break
            pathm[k] |= 1 << b
            pos = 2 * pos + 1 + ((prefr[k] >> b) & 1)
        return chosen, pathm, pos

    def _on_path(self, pathm, bottom, k, lane):
        return 64 * k + lane == bottom or (k < self.kc and (pathm[k] >> lane) & 1)

    def pop_with_last(self, last, end):
        root = self.get(0)
        chosen, pathm, bottom = self.chase(end)
        J = -1
        for k in range(self.ns):
            m = 0
            for lane in range(64):
                i = 64 * k + lane
                if self._on_path(pathm, bottom, k, lane) and i != 0 and last[0] <= self.s[k][lane][0]:
                    m |= 1 << lane
            if m:
                J = 64 * k + m.bit_length() - 1
        if J >= 0:
            for k in range(self.kc):
                for lane in range(64):
                    if (pathm[k] >> lane) & 1 and 64 * k + lane < J:
                        self.s[k][lane] = chosen[k][lane]
        self.set(J if J >= 0 else 0, last)
        return root

    def pop(self):
        last = self.get(self.len - 1)
        self.len -= 1
        return last if self.len == 0 else self.pop_with_last(last, self.len)

    def push_then_pop_full(self, item):   # push + pop on a heap holding CAP entries; entry CAP never materialises
        cap = self.cap
        cnt, j = 0, 1
        while ((cap + 1) >> j) >= 1:
            if item[0] <= self.get(((cap + 1) >> j) - 1)[0]:
                break
            cnt = j
            j += 1
        last = item
        if cnt > 0:
            last = self.get(((cap + 1) >> 1) - 1)
            for j in range(1, cnt):
                self.set(((cap + 1) >> j) - 1, self.get(((cap + 1) >> (j + 1)) - 1))
            self.set(((cap + 1) >> cnt) - 1, item)
        return self.pop_with_last(last, cap)

    def sift_down_range(self, end):
        elt = self.get(0)
        chosen, pathm, bottom = self.chase(end)
        F = -1
        for k in range(self.ns - 1, -1, -1):
            m = 0
            for lane in range(64):
                i = 64 * k + lane
                if self._on_path(pathm, bottom, k, lane) and i != 0 and elt[0] >= self.s[k][lane][0]:
                    m |= 1 << lane
            if m:
                F = 64 * k + ((m & -m).bit_length() - 1)
        dest = (F - 1) >> 1 if F >= 0 else bottom
        for k in range(self.kc):
            for lane in range(64):
                if (pathm[k] >> lane) & 1 and 64 * k + lane < dest:
                    self.s[k][lane] = chosen[k][lane]
        self.set(dest, elt)

    def into_sorted_vec(self):
        end = self.len
        while end > 1:
            end -= 1
            a, b = self.get(0), self.get(end)
            self.set(0, b)
            self.set(end, a)
            self.sift_down_range(end)
        return [self.get(i) for i in range(self.len)]


class MemHeapEmu:
    """candidate_points: flat array; heap_push (cooperative ancestors), heap_push_cached, heap_pop3."""

    def __init__(self):
        self.h = []

    def push(self, item):             # heap_push: all ancestors read at once, one ballot decides how many move
        pos = len(self.h)
        self.h.append(None)
        if pos == 0 or item[0] <= self.h[(pos - 1) >> 1][0]:
            self.h[pos] = item
            return
        anc = []                      # lane j >= 1 holds ancestor j at index ((pos + 1) >> j) - 1
        j = 1
        while ((pos + 1) >> j) >= 1:
            anc.append(((pos + 1) >> j) - 1)
            j += 1
        stop = [j for j, a in enumerate(anc, 1) if item[0] <= self.h[a][0]]
        moved = stop[0] - 1 if stop else len(anc)
        vals = [self.h[a] for a in anc]
        for j in range(1, moved + 1):
            self.h[((pos + 1) >> (j - 1)) - 1] = vals[j - 1]
        self.h[((pos + 1) >> moved) - 1] = item

    def parent_cache(self):
        n = len(self.h)
        pbase = (n - 1) >> 1 if n > 0 else 0
        return pbase, [self.h[pbase + l] if pbase + l < n else None for l in range(64)]

    def push_cached(self, item, pbase, cache):
        pos = len(self.h)
        stays = pos == 0 or item[0] <= cache[((pos - 1) >> 1) - pbase][0]
        if stays:
            self.h.append(item)
            if pos - pbase < 64:
                cache[pos - pbase] = item
            return
        self.push(item)
        for l in range(64):
            cache[l] = self.h[pbase + l] if pbase + l < len(self.h) else None

    def pop3(self):
        last = self.h.pop()
        if not self.h:
            return last
        end = len(self.h)
        root = None
        rounds = []
        p, more = 0, True
        for r in range(3):
            if not more:
                break
            idx, ent, vm = [0] * 64, [None] * 64, 0
            for lane in range(64):
                l1 = lane + 1
                t = l1.bit_length() - 1
                i = ((p + 1) << t) - 1 + (l1 - (1 << t))
                idx[lane] = i
                if lane < 63 and i < end:
                    ent[lane] = self.h[i]
                    vm |= 1 << lane
            if r == 0:
                root = ent[0]
            hm = pm = 0
            for lane in range(31):
                cl = 2 * lane + 1
                if (vm >> cl) & 1:
                    hm |= 1 << lane
                if (vm >> (cl + 1)) & 1 and ent[cl][0] <= ent[cl + 1][0]:
                    pm |= 1 << lane
            cur, bits, more = 0, 0, False
            for step in range(5):
                if not (hm >> cur) & 1:
                    break
                cur = 2 * cur + 1 + ((pm >> cur) & 1)
                bits |= 1 << cur
                if step == 4:
                    more = True
            le = 0
            for lane in range(64):
                if (bits >> lane) & 1 and last[0] <= ent[lane][0]:
                    le |= 1 << lane
            rounds.append((idx, ent, bits, le))
            if more:
                p = idx[cur]
        rstar = max([r for r, x in enumerate(rounds) if x[3]], default=-1)
        if rstar < 0:
            self.h[0] = last
        else:
            writes = []
            for r in range(rstar + 1):
                idx, ent, bits, le = rounds[r]
                lstar = le.bit_length() - 1 if r == rstar else 63
                for lane in range(64):
                    if (bits >> lane) & 1 and lane <= lstar:
                        writes.append(((idx[lane] - 1) >> 1, ent[lane]))
                    if r == rstar and lane == lstar:
                        writes.append((idx[lane], last))
            for i, v in writes:
                self.h[i] = v
        return root


def _script(rng, n_ops, n_keys, p_push, cap=None):
    vals, tags, pops = [], [], []
    size = 0
    for t in range(n_ops):
        do_push = size == 0 or (rng.random() < p_push and (cap is None or size < cap))
        if do_push:
            vals.append(float(rng.integers(0, n_keys)))
            pops.append(0)
            size += 1
        else:
            vals.append(0.0)
            pops.append(1)
            size -= 1
        tags.append(t)
    return np.array(vals, np.float32), np.array(tags, np.int32), np.array(pops, np.uint8)


@pytest.mark.parametrize("ns", [1, 2, 4, 16])
def test_register_heap_matches_the_literal_heap(oracle, ns):
    rng = np.random.default_rng(100 + ns)
    for trial in range(12 if ns < 16 else 4):
        vals, tags, pops = _script(rng, int(rng.integers(50, 900 if ns > 1 else 300)), int(rng.integers(2, 9)), 0.62, cap=64 * ns - 1)
        pv, pt, sv, st = oracle.heap_script(vals, tags, pops)
        h = RegHeapEmu(ns)
        got = []
        for v, t, is_pop in zip(vals, tags, pops):
            if is_pop:
                got.append(h.pop())
            else:
                h.push((float(v), int(t)))
        assert [g[1] for g in got] == pt.tolist()
        assert [g[0] for g in got] == pv.tolist()
        assert [e[1] for e in h.into_sorted_vec()] == st.tolist()


@pytest.mark.parametrize("ns", [1, 2, 4])
def test_fused_push_pop_on_a_full_register_heap(oracle, ns):
    """return_points at ef = 64 NS entries: every further accepted neighbour is push + pop, fused on the device."""
    rng = np.random.default_rng(7 + ns)
    cap = 64 * ns
    for trial in range(6):
        nk = int(rng.integers(3, 12))
        fill = rng.integers(0, nk, cap).astype(np.float32)
        extra = rng.integers(0, nk, 200).astype(np.float32)
        vals, tags, pops = [], [], []
        for i, v in enumerate(fill):
            vals.append(v); tags.append(i); pops.append(0)
        for i, v in enumerate(extra):
            vals.append(v); tags.append(cap + i); pops.append(0)
            vals.append(0.0); tags.append(-1); pops.append(1)
        pv, pt, sv, st = oracle.heap_script(np.array(vals, np.float32), np.array(tags, np.int32), np.array(pops, np.uint8))
        h = RegHeapEmu(ns)
        for i, v in enumerate(fill):
            h.push((float(v), i))
        got = [h.push_then_pop_full((float(v), cap + i)) for i, v in enumerate(extra)]
        assert [g[1] for g in got] == pt.tolist()
        assert [e[1] for e in h.into_sorted_vec()] == st.tolist()


def test_memory_heap_pop3_and_cached_push_match_the_literal_heap(oracle):
    rng = np.random.default_rng(5)
    for trial, n_ops in enumerate([40, 300, 2500, 9000]):
        vals, tags, pops = _script(rng, n_ops, int(rng.integers(2, 10)), 0.7)
        left = int((pops == 0).sum() - (pops == 1).sum())   # then pop everything that is left
        vals = np.concatenate([vals, np.zeros(left, np.float32)])
        tags = np.concatenate([tags, np.full(left, -1, np.int32)])
        pops = np.concatenate([pops, np.ones(left, np.uint8)])
        pv, pt, sv, st = oracle.heap_script(vals, tags, pops)
        assert len(st) == 0
        h = MemHeapEmu()
        got = []
        i = 0
        while i < len(vals):
            if pops[i]:
                got.append(h.pop3())
                i += 1
                continue
            # a batch of consecutive pushes shares one prefetched block of parents, like one expansion on the device
            pbase, cache = h.parent_cache()
            nb = 0
            while i < len(vals) and not pops[i] and nb < 64:
                h.push_cached((float(vals[i]), int(tags[i])), pbase, cache)
                i += 1
                nb += 1
        assert not h.h
        assert [g[1] for g in got] == pt.tolist()
        assert [g[0] for g in got] == pv.tolist()
