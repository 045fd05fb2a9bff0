"""The N > 1 path on CPU (no GPU in the build container): the shard plan, the two-phase exchange — shard lists to the query's home
rank, surviving pairs to their owner rank — and the edge gather run as 2 and 4 `gloo` ranks of tests/dist_model.py, a host-side model
of unicore_amd/csrc/uc_multi.cpp with the oracle standing in for the HIP kernels.  The same data movement through the real library
(device buffers, RCCL or device copies) is covered on the GPU box: tests/test_cli_gpu.py (virtual ranks), tests/test_multi_gpu.py."""
import os
import socket

import numpy as np
import pytest

import util
import unicore_amd as U
import dist_model as ucdist
from oracle import oracle_py as O


def test_shard_ranges_are_partitions_and_the_grid_covers_every_cell_once():
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 2000, 1000)
    for w in (1, 2, 3, 4, 8):
        r = ucdist.shard_ranges(lens, w)
        assert r[0][0] == 0 and r[-1][1] == len(lens) and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        res = [lens[b:e].sum() for b, e in r]
        assert max(res) - min(res) <= 2 * lens.max()        # ~equal residues per shard
    assert ucdist.grid_shape(8) == (1, 8) and ucdist.grid_shape(8, 2) == (4, 2) and ucdist.grid_shape(8, 1) == (8, 1)     # default: T = N, the north-star layout
    with pytest.raises(ValueError):
        ucdist.grid_shape(8, 3)
    big = np.full(80_000, 192)
    for t in (8, 2, 1):
        cover = np.zeros((8, 8), int)                            # every (query octile, target octile) cell exactly once
        for tb, te, qb, qe in ucdist.grid_ranges(big, 8, t):
            cover[qb // 10000: -(-qe // 10000), tb // 10000: -(-te // 10000)] += 1
        assert (cover == 1).all()


def test_pair_owner_puts_mutual_hits_on_one_rank_and_spreads_queries():
    rng = np.random.default_rng(3)
    lens = rng.integers(30, 900, 5000)
    a, b = rng.integers(0, 5000, 20000), rng.integers(0, 5000, 20000)
    for w in (2, 3, 8):
        o1, o2 = ucdist.pair_owner(a, b, lens, w), ucdist.pair_owner(b, a, lens, w)
        assert np.array_equal(o1, o2) and o1.min() >= 0 and o1.max() < w
        share = np.bincount(o1, minlength=w) / len(a)
        assert share.max() < 1.3 / w and share.min() > 0.7 / w


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
        oc, oh, rx = ucdist.exchange_two_phase(c, flat, lens, p.max_seqs)       # the two exchanges
        # accepted edges of the pairs this rank owns (oracle alignment as the stand-in for E5/E6)
        import ctypes
        off = np.concatenate([[0], np.cumsum(oc.astype(np.int64))])
        edges = []
        for q in range(odb.n):
            ms = O.lib().uco_min_score(p, int(lens[q]), int(lens.sum()))
            for k in range(off[q], off[q + 1]):
                a = O.Aln()
                O.lib().uco_align_pair(ctypes.byref(odb.db), q, int(oh["target"][k]), ctypes.byref(p), ms, ctypes.byref(a))
                if a.accepted:
                    edges.append((q, int(oh["target"][k])))
        alle = ucdist.gather_edges(np.array(edges, np.uint32).reshape(-1, 2))
        if rank == 0:
            np.save(os.path.join(tmpdir, "assign.npy"), U.setcover(odb.n, alle))
        np.save(os.path.join(tmpdir, "cnt%d.npy" % rank), oc)
        np.save(os.path.join(tmpdir, "hits%d.npy" % rank), oh)
        np.save(os.path.join(tmpdir, "rx%d.npy" % rank), np.array([rx]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,target_shards", [(2, 0), (2, 1), (4, 2), (4, 0)])      # T = N / two query groups / a Q2 x T2 grid / T = N = 4
def test_gloo_two_phase_exchange_matches_single_rank(tmp_path, world, target_shards):
    """over all ranks every pair of the unsharded lists is owned exactly once, with its score and diagonal; mutual hits sit on one
    rank; the clusters equal the single-rank result"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path), target_shards), nprocs=world, join=True)
    s3, sa = util.family_db(23, n_fam=7, members=5, lmin=40, lmax=120)
    odb = O.OracleDb(s3=s3, sa=sa)
    p = util.oracle_params(O, "-c 0.8 --max-seqs 5")
    ref = O.cluster(odb, p, threads=2)
    want = {}
    for q in range(odb.n):
        for k in range(int(ref["hit_cnt"][q])):
            h = ref["hits"][q, k]
            want[(q, int(h["t"]))] = (int(h["score"]), int(h["diag"]))
    got, owner_of = {}, {}
    for r in range(world):
        oc, oh = np.load(tmp_path / ("cnt%d.npy" % r)), np.load(tmp_path / ("hits%d.npy" % r))
        qs = np.repeat(np.arange(odb.n), oc)
        for q, h in zip(qs.tolist(), oh):
            key = (q, int(h["target"]))
            assert key not in got, key                                   # owned exactly once
            got[key] = (int(h["score"]), int(h["diag"]))
            owner_of[key] = r
        assert int(np.load(tmp_path / ("rx%d.npy" % r))[0]) > 0 or world == 1
    assert got == want
    for (q, t), r in owner_of.items():
        if (t, q) in owner_of:
            assert owner_of[(t, q)] == r                                  # mutual hits meet on one rank
    assert np.array_equal(np.load(tmp_path / "assign.npy"), ref["assign"])     # same clusters as 1 rank


def _gather_main(rank, world, port, tmpdir, limit):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ucdist.GATHER_ALL_LIMIT = limit
        mine = (np.arange(2 * (3 + 5 * rank), dtype=np.uint32) + 1000 * rank).reshape(-1, 2)
        got = ucdist.gather_edges(mine)
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
