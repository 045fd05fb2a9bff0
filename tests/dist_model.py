"""tests/dist_model.py — host-side MODEL of the sharded pass for the CPU tests: one process per rank over torch.distributed
with the `gloo` backend, numpy in place of the device buffers, the oracle in place of the HIP kernels.  TEST INFRASTRUCTURE: the product's
N-GPU path is unicore_amd/csrc/uc_multi.cpp (RCCL called from C); this file restates its data movement so that world_size > 1 runs here,
where there is no GPU:

    rank r : index target shard r % T, match query group r // T against it            (grid_ranges  == uc_multi.cpp grid_cell)
    exchange 1: every query has a HOME rank (contiguous query ranges of ~equal residue counts); a rank's lists are grouped by
             query, so the records of one home are one slice: ragged all-to-all (point-to-point sends here: gloo has no alltoall),
             the home rank merges under (score desc, target asc) and keeps max_seqs     (exchange_two_phase, phase 1)
    exchange 2: surviving pairs go to the rank that OWNS them — a hash of the unordered pair's representative query (the shorter
             sequence, ties: smaller id), so mutual hits meet on one rank               (pair_owner == uc_prefilter.hip pair_owner)
    gather   : accepted edges to rank 0, host set cover there                            (gather_edges)
"""
import numpy as np

import unicore_amd as U

HIT_DTYPE = U.HIT_DTYPE
REC = np.dtype([("query", "<u4"), ("target", "<u4"), ("score", "<i4"), ("diag", "<i4")])


def shard_ranges(lens, parts):
    """Contiguous ranges [b, e) with ~equal residue counts (uc_multi.cpp shard_ranges)."""
    lens = np.asarray(lens, np.int64)
    n = len(lens)
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = int(cum[-1])
    bounds = [0]
    for g in range(1, parts):
        bounds.append(int(np.searchsorted(cum, total * g / parts, side="left")))
    bounds.append(n)
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return [(bounds[g], bounds[g + 1]) for g in range(parts)]


def grid_shape(world, target_shards=0):
    """(Q, T): T = world by default — the north-star layout, one target shard per GPU; any divisor of world otherwise"""
    t = target_shards if target_shards > 0 else world
    if t > world or world % t:
        raise ValueError("target shards (%d) must divide the world size (%d)" % (t, world))
    return world // t, t


def grid_ranges(lens, world, target_shards=0):
    """per rank (tb, te, qb, qe)"""
    q, t = grid_shape(world, target_shards)
    tr, qr = shard_ranges(lens, t), shard_ranges(lens, q)
    return [tr[r % t] + qr[r // t] for r in range(world)]


def pair_owner(a, b, lens, world):
    """owner rank of the unordered pair {a, b} (arrays): the device hash of uc_prefilter.hip, restated"""
    a = np.asarray(a, np.int64); b = np.asarray(b, np.int64)
    la, lb = np.asarray(lens)[a], np.asarray(lens)[b]
    rep = np.where((la < lb) | ((la == lb) & (a < b)), a, b).astype(np.uint64)
    m = np.uint64(0xFFFFFFFF)
    h = (rep * np.uint64(0x9E3779B1)) & m
    h ^= h >> np.uint64(15); h = (h * np.uint64(0x2C1B3C6D)) & m
    h ^= h >> np.uint64(12); h = (h * np.uint64(0x297A2D39)) & m
    h ^= h >> np.uint64(15)
    return (h % np.uint64(world)).astype(np.int64)


def _records(counts, hits):
    q = np.repeat(np.arange(len(counts), dtype=np.uint32), np.asarray(counts, np.int64))
    r = np.zeros(len(hits), REC)
    r["query"], r["target"], r["score"], r["diag"] = q, hits["target"], hits["score"], hits["diag"]
    return r


def _all_to_all(send_parts, group=None):
    """ragged all-to-all of byte buffers by point-to-point messages (sizes first)"""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = torch.tensor([len(p) for p in send_parts], dtype=torch.int64)
    allsz = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(allsz, sizes, group=group)
    recv = [None] * world
    ops, keep = [], []
    for p in range(world):
        n_in = int(allsz[p][rank].item())
        if p == rank:
            recv[p] = np.frombuffer(bytes(send_parts[p]), np.uint8).copy()
            continue
        if len(send_parts[p]):
            t = torch.from_numpy(np.frombuffer(bytes(send_parts[p]), np.uint8).copy())
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, p, group))
        buf = torch.empty(n_in, dtype=torch.uint8)
        recv[p] = buf
        if n_in:
            ops.append(dist.P2POp(dist.irecv, buf, p, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return [r.numpy() if hasattr(r, "numpy") else r for r in recv]


def exchange_two_phase(counts, hits, lens, max_seqs, group=None):
    """this rank's per-shard lists (counts[n], hits grouped by query) -> (counts, hits) of the pairs this rank OWNS after the
    two exchanges, plus the bytes it received from peers"""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(lens)
    rec = _records(counts, hits)
    homes = shard_ranges(lens, world)
    off = np.concatenate([[0], np.cumsum(np.asarray(counts, np.int64))])
    got = _all_to_all([rec[off[b]:off[e]].tobytes() for b, e in homes], group)
    rx = sum(len(g) for p, g in enumerate(got) if p != rank)
    # merge at home: every source slice is one "shard part" of the home's queries
    parts = []
    for g in got:
        r = np.frombuffer(g.tobytes() if hasattr(g, "tobytes") else bytes(g), REC)
        c = np.bincount(r["query"], minlength=n).astype(np.uint32)
        h = np.zeros(len(r), HIT_DTYPE)
        h["target"], h["score"], h["diag"] = r["target"], r["score"], r["diag"]
        parts.append((c, h))
    mc, mh = U.hits_merge(n, max_seqs, parts)
    merged = _records(mc, mh)
    # pairs to their owners (a stable partition keeps the list order inside every owner's segment)
    own = pair_owner(merged["query"], merged["target"], lens, world)
    got2 = _all_to_all([merged[own == p].tobytes() for p in range(world)], group)
    rx += sum(len(g) for p, g in enumerate(got2) if p != rank)
    mine = np.concatenate([np.frombuffer(g.tobytes() if hasattr(g, "tobytes") else bytes(g), REC) for g in got2]) if got2 else np.zeros(0, REC)
    order = np.lexsort((mine["target"], 255 - mine["score"], mine["query"]))     # (query, score desc, target asc): the list order
    mine = mine[order]
    oc = np.bincount(mine["query"], minlength=n).astype(np.uint32)
    oh = np.zeros(len(mine), HIT_DTYPE)
    oh["target"], oh["score"], oh["diag"] = mine["target"], mine["score"], mine["diag"]
    return oc, oh, rx


GATHER_ALL_LIMIT = 256 << 20            # bytes of padded edge buffers up to which every rank receives all lists


def gather_edges(edges, group=None, dst=0):
    """Accepted edges of every rank -> rank `dst` (the one that runs the set cover); the other ranks get an empty array."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    e = np.ascontiguousarray(edges, np.uint32).reshape(-1)
    n = torch.tensor([e.size], dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(x.item()) for x in sizes]
    m = max(max(sizes), 1)
    buf = torch.zeros(m, dtype=torch.int32)
    if e.size:
        buf[: e.size] = torch.from_numpy(e.view(np.int32))
    if 4 * m * world <= GATHER_ALL_LIMIT:
        outs = [torch.empty(m, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(outs, buf, group=group)
    else:
        outs = [torch.empty(m, dtype=torch.int32) for _ in range(world)] if rank == dst else None
        dist.gather(buf, outs, dst=dst, group=group)
    if rank != dst:
        return np.zeros((0, 2), np.uint32)
    return np.concatenate([o[:k].numpy().view(np.uint32).reshape(-1, 2) for o, k in zip(outs, sizes)])
