"""Multi-GPU layer: one process per GPU, graph replicated, query batch split into contiguous shards.

The search path has no data-path collective: every query is an independent read-only traversal
(src/hnsw.rs:1618-1620).  The only exchange is the gather of the answers, done with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests) as ONE
collective per batch: a rank's ids, distances and counts live side by side in one byte buffer
(`PackedAnswers`), which the search writes in place and `all_gather_into_tensor` moves whole -- three
collectives of 0.1-1 MB each would pay the launch latency of a collective three times on a ~1 ms step.
`bench.py --gpus N` and the CPU test of the N>1 path (tests/test_sharding.py) both go through this module.
"""


def shard_bounds(nq, world_size, rank):
    """Contiguous, balanced row block of rank `rank`: sizes differ by at most one, order preserved."""
    base, rem = divmod(nq, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard_rows(nq, world_size):
    return (nq + world_size - 1) // world_size


class PackedAnswers:
    """The answers of one shard in ONE contiguous byte buffer: ids i64[rows][k] | dists f32[rows][k] | counts i32[rows]
    (16-byte padded).  `ids`, `dists`, `counts` are typed views of it: hand their data_ptr() to
    hnswgpu_search_batch_device and the kernel writes the collective's send buffer directly."""

    def __init__(self, rows, k, device):
        import torch
        self.rows, self.k = int(rows), int(k)
        o_d = self.rows * self.k * 8
        o_c = o_d + self.rows * self.k * 4
        self.nbytes = (o_c + self.rows * 4 + 15) // 16 * 16
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.ids, self.dists, self.counts = self.views_of(self.buf, self.rows, self.k)

    @staticmethod
    def views_of(buf, rows, k):
        """Typed views (ids, dists, counts) of one packed shard buffer (a 1-D uint8 tensor)."""
        import torch
        o_d = rows * k * 8
        o_c = o_d + rows * k * 4
        return (buf[:o_d].view(torch.int64).view(rows, k), buf[o_d:o_c].view(torch.float32).view(rows, k),
                buf[o_c:o_c + rows * 4].view(torch.int32))


class AnswerGather:
    """All-gather of the shards' packed answers: one `all_gather_into_tensor` per call.  `coll_device` is where the
    group's collectives run (the rank's GPU for nccl/RCCL, the CPU for gloo); a shard buffer that lives elsewhere is
    copied there first (one copy)."""

    def __init__(self, nq_total, k, world_size, coll_device, group=None):
        import torch
        self.nq, self.k, self.world, self.group = int(nq_total), int(k), int(world_size), group
        self.rows = max_shard_rows(self.nq, self.world)
        self.shard_bytes = PackedAnswers(0, k, "cpu").nbytes if self.rows == 0 else PackedAnswers(self.rows, k, "cpu").nbytes
        self.coll_device = torch.device(coll_device)
        self.recv = torch.empty(self.world * self.shard_bytes, dtype=torch.uint8, device=self.coll_device)

    def gather(self, packed, async_op=False):
        """packed: this rank's PackedAnswers with `rows` == max_shard_rows (shorter shards leave the tail unused).
        async_op: returns the collective's work handle at once (wait() on it before `recv` is read or `packed` rewritten):
        the exchange of one batch then overlaps the search of the next (two PackedAnswers / AnswerGather pairs, alternating)."""
        import torch.distributed as dist
        if packed.rows != self.rows or packed.k != self.k:
            raise ValueError("PackedAnswers of %d x %d rows, the gather was sized for %d x %d" % (packed.rows, packed.k, self.rows, self.k))
        send = packed.buf if packed.buf.device == self.coll_device else packed.buf.to(self.coll_device)
        self._send = send  # (kept alive until the collective is done)
        work = dist.all_gather_into_tensor(self.recv, send, group=self.group, async_op=async_op)
        return work if async_op else self.recv

    def in_input_order(self):
        """(ids [nq,k], dists [nq,k], counts [nq]) of the last gather, shards put back side by side."""
        import torch
        parts = ([], [], [])
        for r in range(self.world):
            lo, hi = shard_bounds(self.nq, self.world, r)
            v = PackedAnswers.views_of(self.recv[r * self.shard_bytes:(r + 1) * self.shard_bytes], self.rows, self.k)
            for p, t in zip(parts, v):
                p.append(t[: hi - lo])
        return tuple(torch.cat(p, dim=0) for p in parts)


class OverlappedExchange:
    """The N > 1 exchange of a stream of batches: two PackedAnswers / AnswerGather pairs alternate, so that the all-gather of
    batch i (asynchronous) overlaps the search of batch i + 1.  Per batch: `pk = buffer(i)` (waits for the exchange that last read
    that buffer, two batches ago) -> the search writes pk.ids / pk.dists / pk.counts -> `exchange(i)`.  `drain()` waits for
    everything in flight; `gathered(i)` = the answers of batch i in input order (valid until batch i + 2 is exchanged).
    Every rank must call exchange() the same number of times in the same order: collectives pair up by order, not by content."""

    def __init__(self, nq_total, k, world_size, device, coll_device, group=None, depth=2):
        self.rows = max_shard_rows(nq_total, world_size)
        self.packs = [PackedAnswers(self.rows, k, device) for _ in range(depth)]
        self.gatherers = [AnswerGather(nq_total, k, world_size, coll_device, group) for _ in range(depth)]
        self.in_flight = [None] * depth

    def buffer(self, i):
        b = i % len(self.packs)
        if self.in_flight[b] is not None:
            self.in_flight[b].wait()
            self.in_flight[b] = None
        return self.packs[b]

    def exchange(self, i, overlap=True):
        b = i % len(self.packs)
        if overlap:
            self.in_flight[b] = self.gatherers[b].gather(self.packs[b], async_op=True)
        else:
            self.gatherers[b].gather(self.packs[b])

    def drain(self):
        for b in range(len(self.in_flight)):
            if self.in_flight[b] is not None:
                self.in_flight[b].wait()
                self.in_flight[b] = None

    def gathered(self, i):
        self.drain()
        return self.gatherers[i % len(self.packs)].in_input_order()

    @property
    def shard_bytes(self):
        return self.gatherers[0].shard_bytes


def gather_answers(local_ids, local_dists, local_counts, nq, group=None):
    """All-gather the per-shard answers into input order (one collective).  Arrays are torch tensors on the group's
    device (shape [nq_local, k] / [nq_local]); shards may differ in length by one row."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    k = local_ids.shape[1]
    g = AnswerGather(nq, k, world, local_ids.device, group)
    p = PackedAnswers(g.rows, k, local_ids.device)
    n = local_ids.shape[0]
    p.ids[:n] = local_ids
    p.dists[:n] = local_dists
    p.counts[:n] = local_counts
    g.gather(p)
    return list(g.in_input_order())  # ids [nq,k], dists [nq,k], counts [nq]


def sharded_parallel_search(search_fn, queries, knbn, ef, group=None):
    """Hnsw::parallel_search over all ranks of `group`.

    search_fn(q_local, knbn, ef) -> (ids[nq_local,k] int64, dists[nq_local,k] f32, counts[nq_local] int32) as
    torch tensors; on the GPU box it wraps hnswgpu_search_batch_device on this rank's replica.  Every rank
    passes the SAME full query matrix (numpy or torch); returns the gathered answers in input order."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nq = queries.shape[0]
    lo, hi = shard_bounds(nq, world, rank)
    ids, dists, counts = search_fn(queries[lo:hi], knbn, ef)
    return gather_answers(ids, dists, counts, nq, group)
