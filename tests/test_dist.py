"""The N>1 path on CPU: shard plan, ragged all-gather over torch.distributed (gloo, world_size 2), host merge.
Per-shard hit lists come from the oracle here (no GPU in this container); the exchange + merge + partition
code is exactly what bench.py runs with backend "nccl" (= RCCL) on the GPUs."""
import os
import socket

import numpy as np
import pytest

import util
import unicore_amd as U
from unicore_amd import dist as ucdist
from oracle import oracle_py as O


def test_shard_and_query_ranges_are_partitions():
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 2000, 1000)
    for w in (1, 2, 3, 4, 8):
        r = ucdist.shard_ranges(lens, w)
        assert r[0][0] == 0 and r[-1][1] == len(lens) and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        res = [lens[b:e].sum() for b, e in r]
        assert max(res) - min(res) <= 2 * lens.max()        # ~equal residues per shard
    cnt = rng.integers(0, 20, 1000).astype(np.uint32)
    hits = np.zeros(int(cnt.sum()), U.HIT_DTYPE)
    hits["target"] = rng.integers(0, 1000, len(hits))
    for w in (1, 2, 4, 8):
        r = ucdist.query_ranges(lens, cnt, hits, w)
        assert r[0][0] == 0 and r[-1][1] == len(lens) and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def test_grid_shape_and_ranges():
    """query groups x target shards: T = 1 while the DB fits one prefilter chunk, target shards where they save whole chunks"""
    small = np.full(1000, 300)                                   # 300 k residues: one chunk
    for w in (1, 2, 4, 8):
        assert ucdist.grid_shape(small, w) == (w, 1)
        g = ucdist.grid_ranges(small, w)
        assert len(g) == w and all(x[:2] == (0, 1000) for x in g)
        assert g[0][2] == 0 and g[-1][3] == 1000 and all(g[i][3] == g[i + 1][2] for i in range(w - 1))
    big = np.full(1_000_000, 192)                                # 192 M residues: two chunks unsharded, one per half
    assert ucdist.grid_shape(big, 8) == (4, 2)                   # cost 2 for T = 1 and T = 2: the larger T wins the tie
    assert ucdist.grid_shape(big, 8, target_shards=8) == (1, 8)
    g = ucdist.grid_ranges(big, 8)
    cover = np.zeros((8, 8), int)                                # every (query octile, target octile) cell exactly once
    for tb, te, qb, qe in g:
        cover[qb // 125000: -(-qe // 125000), tb // 125000: -(-te // 125000)] += 1
    assert (cover == 1).all()
    with pytest.raises(ValueError):
        ucdist.grid_shape(small, 8, target_shards=3)


def test_virtual_shards_merge_equals_unsharded_oracle():
    """G virtual shards processed one after the other and merged == the G=1 list (determinism requirement)"""
    s3, sa = util.family_db(17, n_fam=8, members=5, lmin=40, lmax=140)
    odb = O.OracleDb(s3=s3, sa=sa)
    p = util.oracle_params(O, "-c 0.8 --max-seqs 6")
    lens = np.array([len(x) for x in s3])
    cnt1, h1 = O.prefilter_shard(odb, p)
    flat1 = np.concatenate([h1[q, : cnt1[q]] for q in range(odb.n)]).astype(U.HIT_DTYPE)
    for g in (2, 4, 8):
        parts = []
        for tb, te in ucdist.shard_ranges(lens, g):
            c, h = O.prefilter_shard(odb, p, tb, te)
            parts.append((c, np.concatenate([h[q, : c[q]] for q in range(odb.n)]).astype(U.HIT_DTYPE)))
        mc, mh = U.hits_merge(odb.n, p.max_seqs, parts)
        assert np.array_equal(mc, cnt1)
        assert np.array_equal(mh["target"], flat1["target"]) and np.array_equal(mh["score"], flat1["score"]) and np.array_equal(mh["diag"], flat1["diag"])


def _rank_main(rank, world, port, tmpdir, target_shards):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        s3, sa = util.family_db(23, n_fam=7, members=5, lmin=40, lmax=120)
        odb = O.OracleDb(s3=s3, sa=sa)
        p = util.oracle_params(O, "-c 0.8 --max-seqs 5")
        lens = np.array([len(x) for x in s3])
        tb, te, qb0, qe0 = ucdist.grid_ranges(lens, world, target_shards)[rank]
        c, h = O.prefilter_shard(odb, p, tb, te)                      # this rank's shard (stands in for the HIP prefilter)
        c[:qb0] = 0; c[qe0:] = 0                                      # ... restricted to its query group
        flat = np.concatenate([h[q, : c[q]] for q in range(odb.n)]).astype(U.HIT_DTYPE)
        parts = ucdist.exchange_hits(c, flat, device="cpu")           # the collective
        mc, mh = ucdist.merged_hits(parts, odb.n, p.max_seqs)
        qb, qe = ucdist.query_ranges(lens, mc, mh, world)[rank]
        # accepted edges of this rank's query range (oracle alignment as the stand-in for E5/E6)
        import ctypes
        off = np.concatenate([[0], np.cumsum(mc.astype(np.int64))])
        edges = []
        for q in range(qb, qe):
            ms = O.lib().uco_min_score(p, int(lens[q]), int(lens.sum()))
            for k in range(off[q], off[q + 1]):
                a = O.Aln()
                O.lib().uco_align_pair(ctypes.byref(odb.db), q, int(mh["target"][k]), ctypes.byref(p), ms, ctypes.byref(a))
                if a.accepted:
                    edges.append((q, int(mh["target"][k])))
        alle = ucdist.gather_edges(np.array(edges, np.uint32).reshape(-1, 2), device="cpu")
        np.save(os.path.join(tmpdir, "rank%d.npy" % rank), U.setcover(odb.n, alle) if rank == 0 else mc)
        np.save(os.path.join(tmpdir, "hits%d.npy" % rank), mh)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,target_shards", [(2, 2), (2, 1), (4, 2)])      # two target shards / two query groups / a Q2 x T2 grid
def test_gloo_exchange_matches_single_rank(tmp_path, world, target_shards):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path), target_shards), nprocs=world, join=True)
    s3, sa = util.family_db(23, n_fam=7, members=5, lmin=40, lmax=120)
    odb = O.OracleDb(s3=s3, sa=sa)
    p = util.oracle_params(O, "-c 0.8 --max-seqs 5")
    ref = O.cluster(odb, p, threads=2)
    h0 = np.load(tmp_path / "hits0.npy")
    for r in range(1, world):
        assert np.array_equal(h0, np.load(tmp_path / ("hits%d.npy" % r)))                 # every rank holds the same merged lists
    flat = np.concatenate([ref["hits"][q, : ref["hit_cnt"][q]] for q in range(odb.n)])
    assert np.array_equal(h0["target"], flat["t"]) and np.array_equal(h0["score"], flat["score"])
    assert np.array_equal(np.load(tmp_path / "rank1.npy"), ref["hit_cnt"])
    assert np.array_equal(np.load(tmp_path / "rank0.npy"), ref["assign"])                  # same clusters as 1 rank


def _gather_main(rank, world, port, tmpdir, limit):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ucdist.GATHER_ALL_LIMIT = limit
        mine = (np.arange(2 * (3 + 5 * rank), dtype=np.uint32) + 1000 * rank).reshape(-1, 2)
        got = ucdist.gather_edges(mine, device="cpu")
        np.save(os.path.join(tmpdir, "g%d.npy" % rank), got)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("limit", [1 << 20, 0])          # all-gather path / gather-to-rank-0 path
def test_gather_edges_two_ranks(tmp_path, limit):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gather_main, args=(2, port, str(tmp_path), limit), nprocs=2, join=True)
    want = np.concatenate([(np.arange(2 * (3 + 5 * r), dtype=np.uint32) + 1000 * r).reshape(-1, 2) for r in range(2)])
    assert np.array_equal(np.load(tmp_path / "g0.npy"), want)
    assert len(np.load(tmp_path / "g1.npy")) == 0
