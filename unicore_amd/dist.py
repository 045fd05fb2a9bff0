"""unicore_amd.dist — Python/torch.distributed TEST DRIVER of the multi-GPU layout, one process per rank.

The product's N-GPU path lives in the library (csrc/uc_multi.cpp: RCCL called from C, T = N target shards by default,
`uc_cluster` num_gpus / `uc_comm_*` + `uc_engine_cluster_step`); bench.py and the CLIs use that.  This module remains as the
host-side stand-in the CPU tests run with 2-4 `gloo` ranks (tests/test_dist.py) and keeps the cost-based grid heuristic
(grid_shape) that picked query groups for small databases; it is not on any product path.

    rank r:  index target shard r % T  ->  match query group r // T against it (E1-E4)  ->  per-rank hit lists
             (Q x T = world; grid_shape picks T: 1 while the DB fits one prefilter chunk, see there)
    exchange: all-gather of the ragged hit lists (torch.distributed; backend "nccl" is RCCL over xGMI on
              ROCm, "gloo" on CPU for the tests) — the ONE collective of the path
    every rank: merge the shard lists per query under the frozen order (score desc, target asc), keep
              max_seqs  ->  align its share of the pairs (E5/E6)  ->  accepted edges
              (GPU path: merge + share selection on the device, share = hash of the unordered pair so that
               mutual hits meet on one rank; host path used by the CPU tests: contiguous query ranges)
    gather:   edges to rank 0, which runs the host-side set cover (E7) and writes the result.

The target DB is range-partitioned by residue count; per-shard truncation to max_seqs is lossless because
the global top-M is contained in the union of the shard top-Ms, so the merged result is independent of the
number of shards (tests/test_dist.py checks that with virtual shards and with 2 gloo ranks).
torch is used for device buffers and the collective only.
"""
import numpy as np

from . import HIT_DTYPE, hits_merge


def shard_ranges(lens, world):
    """Contiguous target ranges [b, e) with ~equal residue counts (range partition by key)."""
    lens = np.asarray(lens, np.int64)
    n = len(lens)
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = int(cum[-1])
    bounds = [0]
    for g in range(1, world):
        bounds.append(int(np.searchsorted(cum, total * g / world, side="left")))
    bounds.append(n)
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return [(bounds[g], bounds[g + 1]) for g in range(world)]


PREFILTER_CHUNK_RESIDUES = 96_000_000      # the engine's target chunk (uc_engine.h prefilter_chunk_residues)


def grid_shape(lens, world, target_shards=None):
    """(query groups Q, target shards T) with Q * T == world.

    Rank r matches query group r // T against target shard r % T.  The similar-k-mer generation of a rank is paid once
    per (query of its group) x (target chunk of its shard: the engine walks a shard in chunks of PREFILTER_CHUNK_RESIDUES),
    so the plan minimises  queries_per_rank x chunks_per_shard = ceil(R / T / chunk) * T / world;  ties go to the larger T
    (smaller index per rank).  A database that fits one chunk (BASELINE configs[1]) therefore runs T = 1 — every rank
    indexes all targets (a few ms) and range-partitions the DB on the query side; the target range partition takes over
    where a shard saves whole chunks.  UC_TARGET_SHARDS (or `target_shards`) forces T."""
    import os
    if target_shards is None and os.environ.get("UC_TARGET_SHARDS"):
        target_shards = int(os.environ["UC_TARGET_SHARDS"])
    if target_shards is not None:
        if target_shards < 1 or world % target_shards:
            raise ValueError("target shards (%d) must divide the world size (%d)" % (target_shards, world))
        return world // target_shards, target_shards
    total = int(np.asarray(lens, np.int64).sum())
    best = None
    for t in range(1, world + 1):
        if world % t:
            continue
        chunks = max(1, -(-total // (t * PREFILTER_CHUNK_RESIDUES)))
        cost = chunks * t          # x queries / world, common to all t
        if best is None or cost < best[0] or (cost == best[0] and t > best[1]):
            best = (cost, t)
    return world // best[1], best[1]


def grid_ranges(lens, world, target_shards=None):
    """Per rank (tb, te, qb, qe): target shard and query group, both contiguous ranges with ~equal residue counts."""
    q, t = grid_shape(lens, world, target_shards)
    tr, qr = shard_ranges(lens, t), shard_ranges(lens, q)
    return [tr[r % t] + qr[r // t] for r in range(world)]


def query_ranges(lens, counts, hits, world):
    """Contiguous query ranges with ~equal gapped-DP work (cells = Lq * sum of target lengths)."""
    lens = np.asarray(lens, np.int64)
    counts = np.asarray(counts, np.int64)
    n = len(lens)
    tl = lens[np.asarray(hits["target"], np.int64)] if len(hits) else np.zeros(0, np.int64)
    off = np.concatenate([[0], np.cumsum(counts)])
    csum = np.concatenate([[0], np.cumsum(tl)])
    work = lens * (csum[off[1:]] - csum[off[:-1]])
    cum = np.concatenate([[0], np.cumsum(work)])
    total = int(cum[-1])
    bounds = [0]
    for g in range(1, world):
        bounds.append(int(np.searchsorted(cum, total * g / world, side="left")))
    bounds.append(n)
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return [(bounds[g], bounds[g + 1]) for g in range(world)]


def _allgather_ragged(arr_u8, device, group=None):
    """all-gather of one ragged byte buffer per rank -> list of np.uint8 arrays."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = torch.tensor([arr_u8.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    buf = torch.zeros(m, dtype=torch.uint8, device=device)
    if arr_u8.size:
        buf[: arr_u8.size] = torch.from_numpy(arr_u8).to(device)
    outs = [torch.empty(m, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)          # RCCL all-gather over xGMI when device is a GPU
    return [o[:s].cpu().numpy() for o, s in zip(outs, sizes)]


def exchange_hits(counts, hits, device="cpu", group=None):
    """All-gather every rank's (counts, hits) -> list of per-shard (counts, hits)."""
    c = _allgather_ragged(np.ascontiguousarray(counts, np.uint32).view(np.uint8), device, group)
    h = _allgather_ragged(np.ascontiguousarray(hits, HIT_DTYPE).view(np.uint8), device, group)
    return [(ci.view(np.uint32), hi.view(HIT_DTYPE)) for ci, hi in zip(c, h)]


GATHER_ALL_LIMIT = 256 << 20            # bytes of padded edge buffers up to which every rank receives all lists


def gather_edges(edges, device="cpu", group=None, dst=0):
    """Accepted edges of every rank -> rank `dst` (the one that runs the set cover); the other ranks get an empty array.
    Beyond GATHER_ALL_LIMIT a gather, not an all-gather: at BASELINE configs[2] scale the edge lists are gigabytes and
    only one rank needs them."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    e = np.ascontiguousarray(edges, np.uint32).reshape(-1)
    n = torch.tensor([e.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(x.item()) for x in sizes]
    m = max(max(sizes), 1)
    buf = torch.zeros(m, dtype=torch.int32, device=device)
    if e.size:
        buf[: e.size] = torch.from_numpy(e.view(np.int32)).to(device)
    if 4 * m * world <= GATHER_ALL_LIMIT:      # small lists: the all-gather is the better-tuned collective (and far faster on gloo)
        outs = [torch.empty(m, dtype=torch.int32, device=device) for _ in range(world)]
        dist.all_gather(outs, buf, group=group)
    else:
        outs = [torch.empty(m, dtype=torch.int32, device=device) for _ in range(world)] if rank == dst else None
        dist.gather(buf, outs, dst=dst, group=group)
    if rank != dst:
        return np.zeros((0, 2), np.uint32)
    return np.concatenate([o[:k].cpu().numpy().view(np.uint32).reshape(-1, 2) for o, k in zip(outs, sizes)])


EXCHANGE_ONE_SHOT_LIMIT = 256 << 20     # hit records in the union above which the exchange is merged shard by shard


def _export_hits_tensor(engine, device):
    import torch
    n = engine.hits_size()
    buf = torch.empty((4, max(n, 1)), dtype=torch.int32, device=device)
    if n:
        engine.hits_export_dev(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), buf[3].data_ptr())
    return buf, n


def _import_hits_tensor(engine, t, n, rank, world, device):
    import torch
    t = t.contiguous()
    torch.cuda.synchronize(device)
    return engine.hits_import_dev(n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), rank, world)


def exchange_hits_device(engine, rank, world, device, group=None, one_shot_limit=None):
    """Device-resident exchange: the shard's hit lists go from the engine into one int32 tensor [4, n] on the GPU,
    are all-gathered (RCCL over xGMI with backend "nccl"), and the union is merged, truncated and reduced to the
    pairs this rank owns inside the engine - nothing visits the host but the per-rank sizes.
    Unions beyond `one_shot_limit` records (BASELINE configs[2] scale: up to N x n_seqs x max_seqs) are merged shard by
    shard instead - one broadcast per rank, top-M truncation after every merge - so the peak stays at two lists.
    Returns the number of pairs installed (= gapped alignments of this rank)."""
    import torch
    import torch.distributed as dist

    import os
    limit = int(os.environ.get("UC_EXCHANGE_LIMIT", EXCHANGE_ONE_SHOT_LIMIT)) if one_shot_limit is None else one_shot_limit
    buf, nloc = _export_hits_tensor(engine, device)
    sz = torch.tensor([nloc], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(sz) for _ in range(world)]
    dist.all_gather(sizes, sz, group=group)
    sizes = [int(x.item()) for x in sizes]
    nccl = dist.get_backend(group) == "nccl"
    if sum(sizes) > limit:
        acc, nacc = None, 0
        for r in range(world):
            m = max(sizes[r], 1)
            if r == rank:
                part = buf[:, :m].contiguous()
            else:
                part = torch.empty((4, m), dtype=torch.int32, device=device)
            if nccl:
                dist.broadcast(part, src=r, group=group)
            else:                                                   # gloo (tests / one shared GPU): hop through the host
                h = part.cpu()
                dist.broadcast(h, src=r, group=group)
                part = h.to(device)
            part = part[:, : sizes[r]]
            if acc is None:
                acc, nacc = part, sizes[r]
            elif sizes[r]:
                both = torch.cat([acc[:, :nacc], part], dim=1)
                _import_hits_tensor(engine, both, nacc + sizes[r], 0, 1, device)     # merge + top-M, no ownership filter yet
                del both
                acc, nacc = _export_hits_tensor(engine, device)
        return _import_hits_tensor(engine, acc[:, :nacc], nacc, rank, world, device)
    m = max(max(sizes), 1)
    pad = torch.zeros((4, m), dtype=torch.int32, device=device)
    pad[:, :nloc] = buf[:, :nloc]
    if nccl:
        out = torch.empty((world, 4, m), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(out, pad, group=group)          # RCCL all-gather, GPU to GPU
        parts = [out[r, :, : sizes[r]] for r in range(world)]
    else:                                                           # gloo (tests / one shared GPU): hop through the host
        outs = [torch.empty((4, m), dtype=torch.int32) for _ in range(world)]
        dist.all_gather(outs, pad.cpu(), group=group)
        parts = [outs[r][:, : sizes[r]].to(device) for r in range(world)]
    allh = torch.cat(parts, dim=1)
    return _import_hits_tensor(engine, allh, int(allh.shape[1]), rank, world, device)


def merged_hits(parts, n_seqs, max_seqs):
    return hits_merge(n_seqs, max_seqs, parts)


def cluster_step(engine, lens, rank, world, max_seqs, device="cpu", group=None, setcover=None, gpu_device=None, timing=None):
    """One pass of the sharded hot path on this rank.  Returns (assign or None, n_alignments_this_rank).
    timing: optional dict that receives the wall seconds of the phases on this rank."""
    import time
    from . import setcover as host_setcover

    t = [time.perf_counter()]

    def lap(name):
        t.append(time.perf_counter())
        if timing is not None:
            timing[name] = timing.get(name, 0.0) + t[-1] - t[-2]

    n = len(lens)
    tb, te, qb0, qe0 = grid_ranges(lens, world)[rank]
    engine.prefilter(tb, te, qb0, qe0)
    lap("prefilter")
    if world > 1 and gpu_device is not None:
        # device-resident exchange; every rank then aligns the pairs it owns (unordered-pair hash) over all queries
        qb, qe = 0, n
        n_aln = exchange_hits_device(engine, rank, world, gpu_device, group)
    elif world > 1:
        counts, hits = engine.hits()
        parts = exchange_hits(counts, hits, device, group)
        counts, hits = merged_hits(parts, n, max_seqs)
        engine.set_hits(counts, hits)
        qb, qe = query_ranges(lens, counts, hits, world)[rank]
        n_aln = int(np.asarray(counts[qb:qe], np.int64).sum())
    else:   # single GPU: the hit lists never leave HBM
        qb, qe = 0, n
        n_aln = engine.hits_size()
    lap("exchange")
    engine.align(qb, qe)
    lap("align")
    edges = engine.edges()
    if world > 1:
        edges = gather_edges(edges, device, group)
    lap("edges")
    assign = None
    if rank == 0:
        assign = setcover(n, edges) if setcover else (engine.setcover(edges) if hasattr(engine, "setcover") else host_setcover(n, edges))
    lap("setcover")
    return assign, n_aln
