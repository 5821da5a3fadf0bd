"""The exact visited set of the search kernel (hnswlib-rs_amd/csrc/search_kernels.inc: mix_id / unmix_id, bucket_has, bucket_fill,
cell16_test, cell16_insert, visit_cell16 -- 8-cell buckets of 16-bit cells {valid, bucket displacement, rest of the mixed id}),
restated in Python on an array of 32-bit words and checked against a Python set: never a false positive, never a lost id, `no room`
only when the displacement budget is really exhausted -- with tables small enough that home buckets fill up and ids are displaced,
and with the kernel's split protocol (all lanes of a batch TEST first, ids whose home bucket is full are resolved at once, the others
are INSERTED later from the stale snapshot of their test, through a compare-and-swap that other lanes may have beaten)."""
import random

import pytest

M32 = 0xFFFFFFFF


def mix_id(i, idbits):
    mask = M32 if idbits >= 32 else (1 << idbits) - 1
    sh = (idbits + 1) >> 1
    h = (i * 0x9E3779B1) & mask
    h ^= h >> sh
    h = (h * 0x85EBCA6B) & mask
    h ^= h >> sh
    return h


def unmix_id(h, idbits):
    mask = M32 if idbits >= 32 else (1 << idbits) - 1
    sh = (idbits + 1) >> 1
    h ^= h >> sh
    h = (h * 0xA5CB9243) & mask
    h ^= h >> sh
    h = (h * 0x0E8B2F51) & mask
    return h


def pk_min_u16(a, b):
    return (min(a >> 16, b >> 16) << 16) | min(a & 0xFFFF, b & 0xFFFF)


def bucket_has(v, tag):
    tt = (tag * 0x10001) & M32
    m = pk_min_u16(pk_min_u16(v[0] ^ tt, v[1] ^ tt), pk_min_u16(v[2] ^ tt, v[3] ^ tt))
    return (m & 0xFFFF) == 0 or (m >> 16) == 0


def bucket_fill(v):
    return sum(bin(w & 0x80008000).count("1") for w in v)


def maxbd(restbits):
    return 1 << min(15 - restbits, 6)


class Table:
    def __init__(self, tbits, idbits):
        self.tbits, self.idbits = tbits, idbits
        self.restbits = idbits - (tbits - 3)
        assert 0 <= self.restbits <= 13
        self.words = [0] * (1 << (tbits - 1))          # 2^tbits cells of 16 bits
        self.bmask = (1 << (tbits - 3)) - 1

    def block(self, b):
        return self.words[4 * b:4 * b + 4]

    def cas(self, idx, expect, new):
        old = self.words[idx]
        if old == expect:
            self.words[idx] = new
        return old

    # cell16_test: 1 visited, 0 fresh (home bucket has room and does not hold it), 2 home full, not in it
    def test(self, i):
        h = mix_id(i, self.idbits)
        p = {"bucket": h >> self.restbits, "tag0": 0x8000 | (h & ((1 << self.restbits) - 1))}
        p["v"] = self.block(p["bucket"])
        if bucket_has(p["v"], p["tag0"]):
            return 1, p
        return (2 if bucket_fill(p["v"]) == 8 else 0), p

    # cell16_insert from a (possibly stale) probe: 1 done, 2 no room
    def insert(self, p):
        bucket, bd, v = p["bucket"], 0, list(p["v"])
        while True:
            cnt = bucket_fill(v)
            if cnt == 8:
                bd += 1
                if bd == maxbd(self.restbits):
                    return 2
                bucket = (bucket + 1) & self.bmask
                v = self.block(bucket)
                continue
            k = cnt >> 1
            w32 = v[k]
            tag = p["tag0"] | (bd << self.restbits)
            old = self.cas(bucket * 4 + k, w32, w32 | (tag << ((cnt & 1) * 16)))
            if old == w32:
                return 1
            v[k] = old

    # visit_cell16: 0 already visited, 1 newly marked, 2 no room
    def visit(self, i):
        h = mix_id(i, self.idbits)
        bucket = h >> self.restbits
        tag0 = 0x8000 | (h & ((1 << self.restbits) - 1))
        for bd in range(maxbd(self.restbits)):
            tag = tag0 | (bd << self.restbits)
            v = self.block(bucket)
            while True:
                if bucket_has(v, tag):
                    return 0
                cnt = bucket_fill(v)
                if cnt == 8:
                    break
                k = cnt >> 1
                w32 = v[k]
                if self.cas(bucket * 4 + k, w32, w32 | (tag << ((cnt & 1) * 16))) == w32:
                    return 1
                v = self.block(bucket)
            bucket = (bucket + 1) & self.bmask
        return 2

    def ids_stored(self):
        """decode every valid cell back to the id it stands for (the migration to the HBM bitmap does this)"""
        out = []
        for b in range(self.bmask + 1):
            for w in self.block(b):
                for cell in (w & 0xFFFF, w >> 16):
                    if cell & 0x8000:
                        bd = (cell & 0x7FFF) >> self.restbits
                        rest = cell & ((1 << self.restbits) - 1)
                        home = (b - bd) & self.bmask
                        out.append(unmix_id((home << self.restbits) | rest, self.idbits))
        return out


@pytest.mark.parametrize("idbits", [1, 2, 3, 5, 8, 11, 12, 16, 17, 20, 21, 24, 31, 32])
def test_mix_is_a_bijection_and_unmix_its_inverse(idbits):
    rnd = random.Random(idbits)
    n = 1 << idbits
    sample = range(n) if idbits <= 16 else [rnd.randrange(n) for _ in range(50000)] + [0, 1, n - 1, n - 2]
    seen = set()
    for i in sample:
        h = mix_id(i, idbits)
        assert 0 <= h < n and unmix_id(h, idbits) == i
        seen.add(h)
    if idbits <= 16:
        assert len(seen) == n


@pytest.mark.parametrize("tbits,idbits", [(8, 10), (8, 14), (8, 18), (9, 20), (10, 20), (11, 21), (11, 24), (6, 8), (5, 15)])
def test_table_is_exact_under_the_kernels_batch_protocol(tbits, idbits):
    """batches of up to 64 distinct ids (a neighbour list): test all, resolve `home bucket full` at once, insert the rest from
    their stale probes in a random order.  The table holds at most 2^tbits ids; it is driven until `no room` shows up."""
    if idbits - (tbits - 3) > 13:
        pytest.skip("16-bit cells hold at most 13 rest bits: the host takes 32-bit cells for such a table")
    rnd = random.Random(1000 * tbits + idbits)
    for rep in range(6):
        t = Table(tbits, idbits)
        truth, n_ids, overflowed = set(), 1 << idbits, False
        for _ in range(400):
            batch = rnd.sample(range(n_ids), min(n_ids, rnd.choice([1, 5, 16, 32, 48, 64])))
            if truth and rnd.random() < 0.5:                      # revisit ids that are in already
                old = rnd.sample(sorted(truth), min(len(truth), 10))
                batch = list(dict.fromkeys(old + batch))[:64]
            probes, fresh = {}, []
            for i in batch:                                        # phase 1: every lane tests
                r, p = t.test(i)
                probes[i] = (r, p)
                assert not (r == 1 and i not in truth), "false positive in the home bucket"
                assert not (r == 0 and i in truth), "an id that is in the table looked fresh"
            for i in batch:                                        # phase 2: full home buckets resolved now
                r, p = probes[i]
                if r == 2:
                    vr = t.visit(i)
                    assert (vr == 0) == (i in truth)
                    if vr == 1:
                        truth.add(i)
                    overflowed |= vr == 2
                elif r == 0:
                    fresh.append(i)
            rnd.shuffle(fresh)
            for i in fresh:                                        # phase 3: insertions from stale snapshots
                irc = t.insert(probes[i][1])
                if irc == 1:
                    truth.add(i)
                overflowed |= irc == 2
            if overflowed:
                break
            assert sorted(t.ids_stored()) == sorted(truth)
            if rnd.random() < 0.2:                                 # everything that is in is found, through either entry
                for i in rnd.sample(sorted(truth), min(len(truth), 20)):
                    assert t.test(i)[0] in (1, 2) and t.visit(i) == 0
        # whatever happened, the cells decode to ids that really were inserted (no room => the query moves to the bitmap with them)
        assert set(t.ids_stored()) <= truth | set(batch)
        assert len(t.ids_stored()) == len(set(t.ids_stored()))
        if tbits <= 9:
            assert overflowed or len(truth) > (1 << tbits) * 0.5


def test_no_room_only_when_the_displacement_budget_is_spent():
    """ids crafted into ONE home bucket: 8 stay, the next ones are displaced bucket by bucket; `no room` comes exactly when the
    maxbd buckets from home on are full"""
    tbits, idbits = 9, 18                       # 64 buckets, restbits 12 -> displacement field 3 bits: maxbd 8
    t = Table(tbits, idbits)
    rest = t.restbits
    home = 17
    ids = [unmix_id((home << rest) | r, idbits) for r in range(200)]
    placed = 0
    for i in ids:
        r = t.visit(i)
        if r == 2:
            break
        assert r == 1
        placed += 1
    assert placed == 8 * maxbd(rest)            # 8 buckets of 8 cells
    for i in ids[:placed]:
        assert t.visit(i) == 0
    assert sorted(t.ids_stored()) == sorted(ids[:placed])
    # a neighbouring home bucket now starts displaced, and is still exact
    other = [unmix_id((((home + 1) & t.bmask) << rest) | r, idbits) for r in range(5)]
    for i in other:
        assert t.visit(i) == 1
    for i in other:
        assert t.visit(i) == 0 and t.test(i)[0] == 2
