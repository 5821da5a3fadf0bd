"""Multi-GPU layer: one process per GPU, graph replicated, query batch split into contiguous shards.

The search path has no data-path collective: every query is an independent read-only traversal
(src/hnsw.rs:1618-1620).  The only exchange is the gather of the answers, done with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).
"""
import numpy as np


def shard_bounds(nq, world_size, rank):
    """Contiguous, balanced row block of rank `rank`: sizes differ by at most one, order preserved."""
    base, rem = divmod(nq, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_answers(local_ids, local_dists, local_counts, nq, group=None):
    """All-gather the per-shard answers into input order.  Arrays are torch tensors on the group's device
    (shape [nq_local, k] / [nq_local]); shards may differ in length by one row (padded for the collective)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    k = local_ids.shape[1]
    max_rows = (nq + world - 1) // world

    def pad(t):
        if t.shape[0] == max_rows:
            return t.contiguous()
        out = torch.zeros((max_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        out[: t.shape[0]] = t
        return out

    outs = []
    for t in (local_ids, local_dists, local_counts):
        buf = [torch.empty_like(pad(t)) for _ in range(world)]
        dist.all_gather(buf, pad(t), group=group)
        parts = []
        for r in range(world):
            lo, hi = shard_bounds(nq, world, r)
            parts.append(buf[r][: hi - lo])
        outs.append(torch.cat(parts, dim=0))
    return outs  # ids [nq,k], dists [nq,k], counts [nq]


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
