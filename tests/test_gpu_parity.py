"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs — bit-exact (integer scores, positions, hit lists, cluster assignments, clust.tsv bytes)."""
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle_py
    return oracle_py


@pytest.fixture(scope="module", params=["pk16", "i32"])
def small(O, request):
    """both gapped-kernel variants: packed 16-bit (default; int32 re-run of flagged pairs) and pure int32"""
    s3, sa = util.family_db(11, n_fam=14, members=6, extra=(700, 1100, 1500, 1650, 1780, 2040))
    off, c3, ca = util.flat(s3, sa)
    import unicore_amd as U
    e = U.Engine("-c 0.8 --sw-kernel " + request.param, verbosity=1)
    e.set_db(off, c3, ca)
    return dict(s3=s3, sa=sa, off=off, eng=e, odb=O.OracleDb(s3=s3, sa=sa))


def test_ungapped_parity(O, small):
    rng = np.random.default_rng(5)
    n, N = 4000, len(small["s3"])
    q, t = rng.integers(0, N, n), rng.integers(0, N, n)
    q[:600] = (np.arange(600) // 6) % N          # related pairs (same family) too
    t[:600] = np.minimum(q[:600] // 6 * 6 + rng.integers(0, 6, 600), N - 1)
    lq = np.array([len(small["s3"][i]) for i in q]); lt = np.array([len(small["s3"][i]) for i in t])
    diag = rng.integers(-(lt - 1), np.maximum(lq, 1))
    diag[:600] = rng.integers(-3, 4, 600)
    got = small["eng"].ungapped(q, t, diag)
    p = O.default_params()
    exp = np.array([O.ungapped(small["s3"][a], small["s3"][b], int(d), p) for a, b, d in zip(q, t, diag)])
    assert np.array_equal(got, exp)
    assert exp.max() > 50 and exp.min() == 0   # both regimes exercised


def _pairs(small, n, seed):
    rng = np.random.default_rng(seed)
    N = len(small["s3"])
    q, t = rng.integers(0, N, n), rng.integers(0, N, n)
    k = n // 2
    fam = rng.integers(0, 14, k)
    q[:k] = fam * 6 + rng.integers(0, 6, k)
    t[:k] = fam * 6 + rng.integers(0, 6, k)
    q[k:k + 8] = np.arange(N - 8, N)             # long / tiny / all-X sequences as queries
    t[k + 8:k + 16] = np.arange(N - 8, N)        # ... and as targets
    q[k + 16:k + 24] = np.arange(N - 8, N); t[k + 16:k + 24] = np.arange(N - 8, N)   # self pairs
    return q.astype(np.uint32), t.astype(np.uint32)


@pytest.mark.parametrize("mode", [0, 1])
def test_sw_forward_and_reverse_parity(O, small, mode):
    q, t = _pairs(small, 1500, 21 + mode)
    s, qe, te = small["eng"].sw(mode, q, t)
    p = O.default_params()
    for i in range(len(q)):
        es, eq, et = O.sw(small["s3"][q[i]], small["sa"][q[i]], small["s3"][t[i]], small["sa"][t[i]], p, rev_q=mode)
        assert s[i] == es, (i, q[i], t[i])
        if mode == 0:
            assert (qe[i], te[i]) == (eq, et), (i, q[i], t[i], s[i])


def test_sw_start_pass_parity(O, small):
    q, t = _pairs(small, 1200, 33)
    s, qe, te = small["eng"].sw(0, q, t)
    keep = s > 0
    q, t, s, qe, te = q[keep], t[keep], s[keep], qe[keep], te[keep]
    s2, q2, t2 = small["eng"].sw(2, q, t, qe, te)
    p = O.default_params()
    assert np.array_equal(s2, s)     # reversed prefixes reach the same optimum
    for i in range(len(q)):
        a3, aa = small["s3"][q[i]][: qe[i] + 1], small["sa"][q[i]][: qe[i] + 1]
        b3, ba = small["s3"][t[i]][: te[i] + 1], small["sa"][t[i]][: te[i] + 1]
        es, eq, et = O.sw(a3, aa, b3, ba, p, rev_q=1, rev_t=1)
        assert (s2[i], q2[i], t2[i]) == (es, eq, et), (i, q[i], t[i])


def test_sw_long_query_fallback(O):
    """queries longer than the largest group class (2048 rows) take the generic kernel — same results"""
    rng = np.random.default_rng(9)
    base3, basea = rng.integers(0, 20, 2600, dtype=np.uint8), rng.integers(0, 20, 2600, dtype=np.uint8)
    s3 = [base3, base3[100:2500].copy(), rng.integers(0, 20, 300, dtype=np.uint8), base3[:2100].copy()]
    sa = [basea, basea[100:2500].copy(), rng.integers(0, 20, 300, dtype=np.uint8), basea[:2100].copy()]
    s3[1][::17] = (s3[1][::17] + 3) % 20
    off, c3, ca = util.flat(s3, sa)
    import unicore_amd as U
    e = U.Engine("-c 0.8", verbosity=1)
    e.set_db(off, c3, ca)
    q = np.array([0, 0, 0, 1, 3, 3, 2, 1], np.uint32); t = np.array([1, 2, 0, 0, 0, 2, 0, 3], np.uint32)
    p = O.default_params()
    s, qe, te = e.sw(0, q, t)
    s1, _, _ = e.sw(1, q, t)
    for i in range(len(q)):
        assert (s[i], qe[i], te[i]) == O.sw(s3[q[i]], sa[q[i]], s3[t[i]], sa[t[i]], p), i
        assert s1[i] == O.sw(s3[q[i]], sa[q[i]], s3[t[i]], sa[t[i]], p, rev_q=1)[0], i
    keep = s > 0
    s2, q2, t2 = e.sw(2, q[keep], t[keep], qe[keep], te[keep])
    for k, i in enumerate(np.nonzero(keep)[0]):
        exp = O.sw(s3[q[i]][: qe[i] + 1], sa[q[i]][: qe[i] + 1], s3[t[i]][: te[i] + 1], sa[t[i]][: te[i] + 1], p, rev_q=1, rev_t=1)
        assert (s2[k], q2[k], t2[k]) == exp, i


@pytest.mark.parametrize("opts", ["-c 0.8", "-c 0.5 -s 6 --max-seqs 5", "-c 0.8 --cov-mode 1 -e 1e-6 --rev-correction 0",
                                  "-c 0.5 --min-seq-id 0.3", "-c 0.8 --cov-mode 2 --min-seq-id 0.55",
                                  "-c 0.8 --min-diag-hits 1 --k-score 40", "-c 0.7 --min-diag-hits 3 -s 5",
                                  "-c 0.8 --sym-dedup 0", "-c 0.5 -e 1e-6 --sym-dedup 0 --rev-correction 0",
                                  # optional rule UC-1/M (default off): matrices rescaled by MMseqs2-style bit factors, both sides
                                  "-c 0.8 --mat-bit-factor-3di 2.1 --mat-bit-factor-aa 1.4", "-c 0.5 --mat-bit-factor-3di 2.1 --min-seq-id 0.3 -s 6",
                                  # optional rule UC-1/B (default off): compositional bias on the ungapped score, both sides
                                  "-c 0.8 --comp-bias-corr 1", "-c 0.5 --comp-bias-corr 1 --comp-bias-corr-scale 0.5 --min-seq-id 0.3 --max-seqs 8",
                                  # optional rule UC-1/L (default off): pairs whose lengths alone rule the coverage threshold out are not aligned
                                  # (all-zero record, counted neither as alignment nor in the cells), both sides, every coverage mode
                                  "-c 0.8 --length-gate 1", "-c 0.7 --cov-mode 1 --length-gate 1 --min-seq-id 0.3",
                                  "-c 0.8 --cov-mode 2 --length-gate 1 --sym-dedup 0 -s 6",
                                  # the traceback walk rebuilds every H from a neighbour's and its own byte: neighbours must stay within 127 of each
                                  # other, i.e. largest substitution score + gap open <= 127.  Matrices scaled close to the +-48 limit of either track
                                  # (3Di -48..38, AA -17..48) with the largest gap open: 86 + 31 = 117 (long sequences leave the packed score range and
                                  # take the int32 pass, the others the byte walk)
                                  "-c 0.5 --mat-bit-factor-3di 9.5 --mat-bit-factor-aa 8.7 --gap-open 31 --gap-extend 3 --min-seq-id 0.3",
                                  "-c 0.5 --mat-bit-factor-aa 8.7 --gap-open 31 --gap-extend 31 --min-seq-id 0.2 --max-seqs 50"])
def test_pipeline_stage_parity(O, small, opts):
    """prefilter hit lists, per-pair alignment records, edges and the set cover all equal the oracle's"""
    import unicore_amd as U
    e = U.Engine(opts, verbosity=1)
    e.set_db(small["off"], *util.flat(small["s3"], small["sa"])[1:])
    p = util.oracle_params(O, opts)
    ref = O.cluster(small["odb"], p, threads=8)
    e.prefilter()
    cnt, hits = e.hits()
    assert np.array_equal(cnt, ref["hit_cnt"])
    rh = np.concatenate([ref["hits"][i, : cnt[i]] for i in range(len(cnt))])
    assert np.array_equal(hits["target"], rh["t"]) and np.array_equal(hits["score"], rh["score"]) and np.array_equal(hits["diag"], rh["diag"])
    e.align()
    al = e.alns()
    ra = np.concatenate([ref["aln"][i, : cnt[i]] for i in range(len(cnt))])
    for f in ("score", "score_rev", "corrected", "pass_evalue", "accepted"):
        assert np.array_equal(al[f], ra[f]), f
    pe = al["pass_evalue"] == 1
    for f in ("qstart", "qend", "tstart", "tend"):
        assert np.array_equal(al[f][pe], ra[f][pe]), f
    for f in ("aln_len", "idents"):          # traceback statistics (only computed under --min-seq-id)
        assert np.array_equal(al[f], ra[f]), f
    if "--min-seq-id" in opts:
        assert (ra["aln_len"] > 0).sum() > 50
        if "0.55" in opts:
            assert 0 < ra["accepted"][ra["aln_len"] > 0].mean() < 1   # the identity gate bites both ways
    st = e.stats()
    c = ref["counts"]
    if "--length-gate 1" in opts:             # the gate bites, and not on everything; what it lets through has the records of the rule-off run
        assert 0 < c["n_alignments"] < int(cnt.sum()) and c["n_edges"] > 0
        off_ref = O.cluster(small["odb"], util.oracle_params(O, opts.replace("--length-gate 1", "--length-gate 0")), threads=8)
        oa = np.concatenate([off_ref["aln"][i, : cnt[i]] for i in range(len(cnt))])
        lens = np.diff(small["off"]).astype(np.float32)
        ql, tl = lens[np.repeat(np.arange(len(cnt)), cnt)], lens[hits["target"]]
        cov = np.float32(p.cov)
        keep = ((ql / tl >= cov) & (tl / ql >= cov)) if p.cov_mode == 0 else (ql / tl >= cov) if p.cov_mode == 1 else (tl / ql >= cov)
        assert int(keep.sum()) == c["n_alignments"]
        assert ra[keep].tobytes() == oa[keep].tobytes() and not ra[~keep].view(np.uint8).any()
    for a, b in (("n_sim_kmers", "n_sim_kmers"), ("n_kmer_hits", "n_kmer_hits"), ("n_candidates", "n_candidates"),
                 ("n_prefilter_hits", "n_prefilter_hits"), ("n_gapped_alignments", "n_alignments"), ("n_edges", "n_edges"),
                 ("cells_fwd", "cells_fwd"), ("cells_start", "cells_start")):
        assert st[a] == c[b], (a, st[a], c[b])
    assign = U.setcover(e.n, e.edges())
    assert np.array_equal(assign, ref["assign"])
    assert np.array_equal(e.setcover(e.edges()), ref["assign"])          # graph built on the GPU, greedy cover on the host
    assert len(set(assign.tolist())) < e.n   # something actually clustered
    # the int32-only kernel path gives the same records
    e2 = U.Engine(opts + " --sw-kernel i32", verbosity=1)
    e2.set_db(small["off"], *util.flat(small["s3"], small["sa"])[1:])
    e2.set_hits(cnt, hits)
    e2.align()
    assert e2.alns().tobytes() == al.tobytes()


def test_cluster_end_to_end_tsv_bytes(O, tmp_path):
    """uc_cluster + uc_createtsv + uc_rmdb on DB files == oracle clust.tsv, byte for byte"""
    import unicore_amd as U
    db = util.gen_synth_db(str(tmp_path / "db"), 6, 0x5EED0001, 48, 0.6)
    out = str(tmp_path / "clu" / "clust")
    os.makedirs(os.path.dirname(out))
    st = U.cluster(db, out + "_cluster", str(tmp_path / "tmp"), "-c 0.8 --single-step-clustering", threads=4)
    U.createtsv(db, out + "_cluster", out + ".tsv")
    odb = O.OracleDb(db)
    ref = O.cluster(odb, util.oracle_params(O, "-c 0.8"), threads=8, dumps=False)
    O.write_tsv(str(tmp_path / "ref.tsv"), odb, ref["assign"])
    assert open(out + ".tsv", "rb").read() == open(str(tmp_path / "ref.tsv"), "rb").read()
    util.tsv_invariants(out + ".tsv", odb.names())
    assert st["n_clusters"] == ref["counts"]["n_clusters"] and st["n_gapped_alignments"] == ref["counts"]["n_alignments"]
    U.rmdb(out + "_cluster")
    assert not os.path.exists(out + "_cluster") and not os.path.exists(out + "_cluster.index")


@pytest.mark.parametrize("opts,steps,sens", [("-c 0.8 --cluster-steps 3", 3, 4.0), ("-c 0.5 -s 6 --cluster-steps 2 --max-seqs 20", 2, 6.0)])
def test_cascade_end_to_end_tsv_bytes(O, tmp_path, opts, steps, sens):
    """E8: rounds on representatives with rising sensitivity + merge == the oracle's cascade"""
    import unicore_amd as U
    db = util.gen_synth_db(str(tmp_path / "db"), 6, 0x5EED0003, 40, 0.6)
    out = str(tmp_path / "clust")
    st = U.cluster(db, out + "_cluster", str(tmp_path / "tmp"), opts + " --linclust 0", threads=4)
    U.createtsv(db, out + "_cluster", out + ".tsv")
    odb = O.OracleDb(db)
    p = util.oracle_params(O, opts.replace(" --cluster-steps %d" % steps, ""))
    thr = O.cascade_thresholds(p, sens, steps)
    assert thr[-1] == p.kmer_thr and thr[0] > thr[-1]
    ref = O.cluster_cascade(odb, p, thr, threads=8)
    O.write_tsv(str(tmp_path / "ref.tsv"), odb, ref["assign"])
    assert open(out + ".tsv", "rb").read() == open(str(tmp_path / "ref.tsv"), "rb").read()
    util.tsv_invariants(out + ".tsv", odb.names())
    assert st["n_clusters"] == ref["counts"]["n_clusters"] and st["n_gapped_alignments"] == ref["counts"]["n_alignments"]
    single = O.cluster(odb, p, threads=8, dumps=False)
    assert ref["round_sizes"][1] < odb.n                                  # the first round removed something
    assert ref["counts"]["n_alignments"] != single["counts"]["n_alignments"]   # and the cascade is not the single step


def test_sw_long_query_row_blocks(O):
    """queries of several 2048-row blocks through the row-blocked kernel (uc_sw_long.hip): boundaries between blocks,
    optimum in a later block, start pass whose masked prefix skips whole blocks"""
    rng = np.random.default_rng(31)
    base3, basea = rng.integers(0, 20, 6500, dtype=np.uint8), rng.integers(0, 20, 6500, dtype=np.uint8)
    s3 = [base3, base3[300:6400].copy(), base3[:4200].copy(), rng.integers(0, 20, 300, dtype=np.uint8), base3[2000:4500].copy(),
          np.concatenate([base3[5000:6400], base3[100:900]])]
    sa = [basea, basea[300:6400].copy(), basea[:4200].copy(), rng.integers(0, 20, 300, dtype=np.uint8), basea[2000:4500].copy(),
          np.concatenate([basea[5000:6400], basea[100:900]])]
    for k in (1, 2, 4):
        mut = rng.random(len(s3[k])) < 0.15
        s3[k][mut] = rng.integers(0, 20, int(mut.sum()), dtype=np.uint8)
        cut = int(rng.integers(500, len(s3[k]) - 500))           # one deletion: the path crosses block boundaries in a gap state too
        s3[k] = np.delete(s3[k], slice(cut, cut + 7)); sa[k] = np.delete(sa[k], slice(cut, cut + 7))
    off, c3, ca = util.flat(s3, sa)
    import unicore_amd as U
    e = U.Engine("-c 0.8", verbosity=1)
    e.set_db(off, c3, ca)
    q = np.array([0, 0, 0, 0, 0, 1, 2, 4, 1, 2, 4, 1, 0], np.uint32); t = np.array([1, 2, 3, 4, 5, 0, 0, 0, 2, 1, 1, 3, 0], np.uint32)
    p = O.default_params()
    s, qe, te = e.sw(0, q, t)
    s1, _, _ = e.sw(1, q, t)
    for i in range(len(q)):
        assert (s[i], qe[i], te[i]) == O.sw(s3[q[i]], sa[q[i]], s3[t[i]], sa[t[i]], p), i
        assert s1[i] == O.sw(s3[q[i]], sa[q[i]], s3[t[i]], sa[t[i]], p, rev_q=1)[0], i
    assert (qe > 4096).any() and ((qe > 2048) & (qe < 4096)).any()
    keep = s > 0
    s2, q2, t2 = e.sw(2, q[keep], t[keep], qe[keep], te[keep])
    lens = np.array([len(x) for x in s3])
    assert ((lens[q[keep]] - 1 - qe[keep]) >= 2048).any()         # a start pass that skips at least one whole block
    for k, i in enumerate(np.nonzero(keep)[0]):
        exp = O.sw(s3[q[i]][: qe[i] + 1], sa[q[i]][: qe[i] + 1], s3[t[i]][: te[i] + 1], sa[t[i]][: te[i] + 1], p, rev_q=1, rev_t=1)
        assert (s2[k], q2[k], t2[k]) == exp, i


def test_long_query_pipeline_with_seqid(O):
    """queries beyond the largest systolic class go through the generic kernel in every pass, including the
    traceback-statistics pass of --min-seq-id"""
    rng = np.random.default_rng(19)
    base3, basea = rng.integers(0, 20, 2500, dtype=np.uint8), rng.integers(0, 20, 2500, dtype=np.uint8)
    s3, sa = [], []
    for m in range(5):
        keep = rng.random(2500) > 0.03
        a3, aa = base3[keep].copy(), basea[keep].copy()
        mut = rng.random(len(a3)) < 0.2
        a3[mut] = rng.integers(0, 20, int(mut.sum()), dtype=np.uint8)
        muta = rng.random(len(aa)) < 0.45
        aa[muta] = rng.integers(0, 20, int(muta.sum()), dtype=np.uint8)
        s3.append(a3); sa.append(aa)
    s3.append(base3[:900].copy()); sa.append(basea[:900].copy())
    off, c3, ca = util.flat(s3, sa)
    import unicore_amd as U
    opts = "-c 0.3 --min-seq-id 0.5"
    e = U.Engine(opts, verbosity=1)
    e.set_db(off, c3, ca)
    e.prefilter(); e.align()
    ref = O.cluster(O.OracleDb(s3=s3, sa=sa), util.oracle_params(O, opts), threads=4)
    cnt, hits = e.hits()
    assert np.array_equal(cnt, ref["hit_cnt"])
    al = e.alns()
    ra = np.concatenate([ref["aln"][i, : cnt[i]] for i in range(len(cnt))])
    for f in ("score", "score_rev", "pass_evalue", "accepted", "aln_len", "idents"):
        assert np.array_equal(al[f], ra[f]), f
    assert (ra["aln_len"] > 2000).any() and 0 < ra["accepted"].mean() < 1
    assert np.array_equal(U.setcover(e.n, e.edges()), ref["assign"])


def test_search_long_queries_m8_bytes(O, tmp_path):
    """search path on sequences of several row blocks: both traceback-statistics passes (alignment length / identities and
    the gap count) run through the row-blocked kernel; the BLAST-tab file equals the oracle's byte for byte"""
    import unicore_amd as U
    rng = np.random.default_rng(23)
    base3, basea = rng.integers(0, 20, 5200, dtype=np.uint8), rng.integers(0, 20, 5200, dtype=np.uint8)
    s3, sa = [], []
    for m in range(5):
        lo, hi = int(rng.integers(0, 400)), int(rng.integers(4700, 5200))
        a3, aa = base3[lo:hi].copy(), basea[lo:hi].copy()
        for _ in range(3):                                    # a few indels
            cut = int(rng.integers(300, len(a3) - 300)); w = int(rng.integers(1, 9))
            a3 = np.delete(a3, slice(cut, cut + w)); aa = np.delete(aa, slice(cut, cut + w))
        mut = rng.random(len(a3)) < 0.15
        a3[mut] = rng.integers(0, 20, int(mut.sum()), dtype=np.uint8)
        muta = rng.random(len(aa)) < 0.3
        aa[muta] = rng.integers(0, 20, int(muta.sum()), dtype=np.uint8)
        s3.append(a3); sa.append(aa)
    s3.append(base3[1000:1900].copy()); sa.append(basea[1000:1900].copy())
    qdbp, tdbp = str(tmp_path / "q"), str(tmp_path / "t")
    util.write_db(qdbp, s3[:3], sa[:3], ["q_%d" % i for i in range(3)])
    util.write_db(tdbp, s3[2:], sa[2:], ["t_%d" % i for i in range(4)])
    out = str(tmp_path / "res")
    opts = "-c 0.5"
    U.search(qdbp, tdbp, out + "_aln", str(tmp_path / "tmp"), opts, threads=4)
    U.convertalis(qdbp, tdbp, out + "_aln", out + ".m8")
    qdb, tdb = O.OracleDb(qdbp), O.OracleDb(tdbp)
    po = util.oracle_params(O, "-e 10 --max-seqs 1000 " + opts)
    O.write_m8(str(tmp_path / "ref.m8"), qdb, tdb, po, O.search(qdb, tdb, po, threads=8))
    got, exp = open(out + ".m8", "rb").read(), open(str(tmp_path / "ref.m8"), "rb").read()
    assert got == exp
    rows = [l.split("\t") for l in got.decode().splitlines()]
    assert len(rows) >= 8 and any(int(r[3]) > 4096 for r in rows) and any(int(r[5]) > 0 for r in rows)


@pytest.mark.parametrize("target_shards", [3, 1, 2])
def test_device_exchange_virtual_ranks(O, target_shards):
    """multi-GPU layout on one GPU: W virtual ranks (query groups x target shards) -> device-side export, union, merge (== unsharded lists),
    pair-hash partition over W virtual ranks: every pair is aligned exactly once, mutual hits meet on one rank, and
    the union of the ranks' accepted edges equals the single-GPU edge set."""
    import torch
    import unicore_amd as U
    import dist_model as ucdist
    s3, sa = util.family_db(29, n_fam=12, members=6, lmin=60, lmax=260)
    off, c3, ca = util.flat(s3, sa)
    lens = np.array([len(x) for x in s3])
    opts = "-c 0.8 --max-seqs 7"
    e = U.Engine(opts, verbosity=1)
    e.set_db(off, c3, ca)
    e.prefilter()
    cnt_ref, hits_ref = e.hits()
    e.align()
    edges_ref = set(map(tuple, e.edges().tolist()))
    assert len(edges_ref) > 50
    W = 3 if target_shards != 2 else 4
    bufs = []
    assert ucdist.grid_shape(W, target_shards) == (W // target_shards, target_shards)
    for tb, te, qb, qe in ucdist.grid_ranges(lens, W, target_shards):
        e.prefilter(tb, te, qb, qe)
        n = e.hits_size()
        t = torch.empty((4, max(n, 1)), dtype=torch.int32, device="cuda")
        e.hits_export_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr())
        bufs.append(t[:, :n])
    allh = torch.cat(bufs, dim=1).contiguous()
    torch.cuda.synchronize()
    ptrs = [allh[i].data_ptr() for i in range(4)]
    ntot = int(allh.shape[1])
    assert ntot >= len(hits_ref)                      # per-shard top-M lists: the union is a superset
    if target_shards == 1:
        assert ntot == len(hits_ref)                  # query groups alone: the rank lists partition the unsharded lists
    assert e.hits_import_dev(ntot, *ptrs, 0, 1) == len(hits_ref)
    cnt, hits = e.hits()
    assert np.array_equal(cnt, cnt_ref) and hits.tobytes() == hits_ref.tobytes()
    total, edges, seen = 0, set(), set()
    for r in range(W):
        k = e.hits_import_dev(ntot, *ptrs, r, W)
        total += k
        cnt, hits = e.hits()
        q = np.repeat(np.arange(len(cnt)), cnt)
        pairs = set(zip(q.tolist(), hits["target"].tolist()))
        assert not (pairs & seen)
        seen |= pairs
        for (a, b) in pairs:                          # a mutual hit is owned by the same rank
            if (b, a) in ref_pairs(cnt_ref, hits_ref):
                assert (b, a) in pairs
        e.align()
        edges |= set(map(tuple, e.edges().tolist()))
    assert total == len(hits_ref) and seen == ref_pairs(cnt_ref, hits_ref)
    assert edges == edges_ref


_REF_PAIRS = {}


def ref_pairs(cnt, hits):
    key = (cnt.tobytes(), hits.tobytes())
    if key not in _REF_PAIRS:
        q = np.repeat(np.arange(len(cnt)), cnt)
        _REF_PAIRS[key] = set(zip(q.tolist(), hits["target"].tolist()))
    return _REF_PAIRS[key]


@pytest.mark.parametrize("opts", ["-c 0.8", "-c 0.3 -e 1e-3 --max-seqs 12 --cov-mode 2", "-c 0.7 --cov-mode 1 --length-gate 1"])      # last: rule UC-1/L on the search path
def test_search_m8_bytes(O, tmp_path, opts):
    """uc_search + uc_convertalis (query DB vs target DB, same kernels, traceback statistics for
    every accepted pair incl. the gap count) == the oracle's search + BLAST-tab writer, byte for byte"""
    import unicore_amd as U
    s3, sa = util.family_db(41, n_fam=10, members=7, lmin=40, lmax=300, extra=(900,))
    order = np.random.default_rng(3).permutation(len(s3))
    qi, ti = sorted(order[:25].tolist()), sorted(order[25:].tolist())
    qdbp, tdbp = str(tmp_path / "q"), str(tmp_path / "t")
    qn = util.write_db(qdbp, [s3[i] for i in qi], [sa[i] for i in qi], ["q_%03d" % i for i in qi])
    tn = util.write_db(tdbp, [s3[i] for i in ti], [sa[i] for i in ti], ["t_%03d" % i for i in ti])
    out = str(tmp_path / "res")
    st = U.search(qdbp, tdbp, out + "_aln", str(tmp_path / "tmp"), opts, threads=4)
    U.convertalis(qdbp, tdbp, out + "_aln", out + ".m8")
    qdb, tdb = O.OracleDb(qdbp), O.OracleDb(tdbp)
    po = util.oracle_params(O, ("-e 10 --max-seqs 1000 " + opts))           # `search` defaults, then the given flags
    ref = O.search(qdb, tdb, po, threads=8)
    O.write_m8(str(tmp_path / "ref.m8"), qdb, tdb, po, ref)
    got, exp = open(out + ".m8", "rb").read(), open(str(tmp_path / "ref.m8"), "rb").read()
    assert got == exp
    rows = [l.split("\t") for l in got.decode().splitlines()]
    assert len(rows) > 30 and all(len(r) == 12 for r in rows)
    assert any(int(r[5]) > 0 for r in rows) and any(int(r[4]) > 0 for r in rows)      # gaps and mismatches occur
    assert all(r[0].startswith("q_") and r[1].startswith("t_") for r in rows)
    qs = [r[0] for r in rows]
    assert all(qs[i] == qs[i - 1] or qs[i] not in qs[:i] for i in range(1, len(qs)))   # rows grouped by query (profile.rs:50-55)
    assert st["n_gapped_alignments"] == ref["counts"]["n_alignments"]
    U.rmdb(out + "_aln")
    assert not os.path.exists(out + "_aln")
    if opts == "-c 0.8":     # the module surface: `unicore search INPUT TARGET OUTPUT TMP` hands TARGET to the engine as the query DB
        import subprocess    # (search.rs:45-46), writes OUTPUT.m8, removes OUTPUT_aln, checkpoint "1"
        exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "unicore")
        cout = str(tmp_path / "cli" / "hits")
        r = subprocess.run([exe, "search", tdbp, qdbp, cout, str(tmp_path / "tmp2"), "-s", opts, "-v", "1"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(cout + ".m8", "rb").read() == exp and not os.path.exists(cout + "_aln")
        assert open(str(tmp_path / "cli" / "search.chk")).read() == "1"


def test_chunked_prefilter_equals_unchunked(O, small):
    """large target ranges are indexed in chunks whose per-query lists are merged on the device: same lists, same
    records (forced here with a tiny chunk size)"""
    e = small["eng"]
    e.prefilter()
    cnt0, hits0 = e.hits()
    st0 = e.stats()
    os.environ["UC_PREFILTER_CHUNK_RES"] = "3000"
    try:
        e.prefilter()
        cnt1, hits1 = e.hits()
        # a sub-range of targets and of queries too (the search path's shape)
        e.prefilter(5, 60, 10, 80)
        cnt2, hits2 = e.hits()
    finally:
        del os.environ["UC_PREFILTER_CHUNK_RES"]
    assert np.array_equal(cnt0, cnt1) and hits0.tobytes() == hits1.tobytes()
    # the chunk x chunk grid is walked as its upper triangle (symmetric hit relation); the full grid gives the same lists and counters
    os.environ["UC_PREFILTER_CHUNK_RES"] = "3000"
    try:
        e.reset_stats(); e.prefilter(); sa = e.stats(); ca, ha = e.hits()
        os.environ["UC_PREFILTER_SYMMETRIC"] = "0"
        e.reset_stats(); e.prefilter(); sb = e.stats(); cb, hb = e.hits()
    finally:
        os.environ.pop("UC_PREFILTER_SYMMETRIC", None)
        del os.environ["UC_PREFILTER_CHUNK_RES"]
    assert np.array_equal(ca, cb) and ha.tobytes() == hb.tobytes() == hits0.tobytes()
    for k in ("n_sim_kmers", "n_kmer_hits", "n_candidates", "n_prefilter_hits"):
        assert sa[k] == sb[k], k
    assert sa["n_filtered_hits"] < sb["n_filtered_hits"]
    e.prefilter(5, 60, 10, 80)
    cnt3, hits3 = e.hits()
    assert np.array_equal(cnt2, cnt3) and hits2.tobytes() == hits3.tobytes()
    assert cnt3[:10].sum() == 0 and cnt3[80:].sum() == 0 and cnt3.sum() > 0 and (hits3["target"] >= 5).all() and (hits3["target"] < 60).all()
    st1 = e.stats()
    assert st1["n_candidates"] - st0["n_candidates"] > 0


def test_prefilter_wide_and_compact_key_forms_agree(O, small):
    """index entries and the per-query (target, diagonal) keys of the double-hit filter are u32 when target and diagonal
    fit 32 bits together (compact, the common case) and u64 otherwise (wide; forced here with UC_PREFILTER_WIDE): same
    hit lists and the same stage counters, whole DB, target sub-range (relative target ids) and chunked"""
    e = small["eng"]
    res = []
    for wide in (False, True):
        if wide:
            os.environ["UC_PREFILTER_WIDE"] = "1"
        try:
            e.reset_stats()
            e.prefilter()
            c0, h0 = e.hits()
            st = e.stats()
            e.prefilter(7, 70, 3, 90)
            c1, h1 = e.hits()
            os.environ["UC_PREFILTER_CHUNK_RES"] = "2500"
            e.prefilter()
            c2, h2 = e.hits()
        finally:
            os.environ.pop("UC_PREFILTER_WIDE", None)
            os.environ.pop("UC_PREFILTER_CHUNK_RES", None)
        res.append((c0.tobytes(), h0.tobytes(), c1.tobytes(), h1.tobytes(), c2.tobytes(), h2.tobytes(),
                    st["n_sim_kmers"], st["n_kmer_hits"], st["n_candidates"], st["n_filtered_hits"]))
    assert res[0] == res[1]
    assert res[0][0] == res[0][4] and res[0][1] == res[0][5]           # chunked == unchunked


def _scaled_matrix(src, dst, factor):
    out = []
    for line in open(src):
        tok = line.split()
        if line.startswith("#") or not tok or not tok[0].isalpha() or len(tok) < 3 or not tok[1].lstrip("-").isdigit():
            out.append(line)
        else:
            out.append(tok[0] + " " + " ".join(str(max(-48, min(48, int(v) * factor))) for v in tok[1:]) + "\n")
    open(dst, "w").writelines(out)


def test_packed_kernel_overflow_is_rerun_in_int32(O, tmp_path):
    """scores beyond the packed kernel's 16-bit range (0x7C00: H = max3 runs on f16 bit patterns) are detected and
    recomputed exactly by the int32 kernel - forced with 4x-scaled matrices on near-identical 1400-residue pairs"""
    import ctypes
    import unicore_amd as U
    m3, ma = str(tmp_path / "m3.out"), str(tmp_path / "ma.out")
    _scaled_matrix(os.path.join(util.ROOT, "unicore_amd", "data", "mat3di_synthetic.out"), m3, 4)
    _scaled_matrix(os.path.join(util.ROOT, "unicore_amd", "data", "blosum62.out"), ma, 4)
    rng = np.random.default_rng(77)
    s3, sa = [], []
    for L in (1400, 1100, 700, 300):
        b3, ba = rng.integers(0, 20, L, dtype=np.uint8), rng.integers(0, 20, L, dtype=np.uint8)
        for _ in range(3):
            a3, aa = b3.copy(), ba.copy()
            mut = rng.random(L) < 0.03
            a3[mut] = rng.integers(0, 20, int(mut.sum()), dtype=np.uint8)
            s3.append(a3); sa.append(aa)
    off, c3, ca = util.flat(s3, sa)
    e = U.Engine("-c 0.8 --mat3di %s --mat-aa %s" % (m3, ma), verbosity=1)
    e.set_db(off, c3, ca)
    p = O.default_params()
    assert O.lib().uco_load_matrix(m3.encode(), p.S3) == 0 and O.lib().uco_load_matrix(ma.encode(), p.SA) == 0
    N = len(s3)
    q, t = np.repeat(np.arange(N), N), np.tile(np.arange(N), N)
    st0 = e.stats()["n_pk_reruns"]
    for mode in (0, 1):
        s, qe, te = e.sw(mode, q, t)
        for i in range(len(q)):
            exp = O.sw(s3[q[i]], sa[q[i]], s3[t[i]], sa[t[i]], p, rev_q=mode)
            assert s[i] == exp[0], (mode, i)
            if mode == 0:
                assert (qe[i], te[i]) == exp[1:], (mode, i)
    assert s.max() >= 0 and e.sw(0, q, t)[0].max() > 0x7C00         # the range was actually exceeded
    assert e.stats()["n_pk_reruns"] > st0


def test_degenerate_databases(O, tmp_path):
    """one sequence, two unrelated short sequences (no k-mer hit at all), a sequence shorter than the k-mer span:
    the pipeline, the cascade and the search path all run through and agree with the oracle"""
    import unicore_amd as U
    rng = np.random.default_rng(123)
    cases = {
        "one": ([rng.integers(0, 20, 80, dtype=np.uint8)], [rng.integers(0, 20, 80, dtype=np.uint8)]),
        "tiny": ([rng.integers(0, 20, 4, dtype=np.uint8), rng.integers(0, 20, 7, dtype=np.uint8)],
                 [rng.integers(0, 20, 4, dtype=np.uint8), rng.integers(0, 20, 7, dtype=np.uint8)]),
        "unrelated": ([rng.integers(0, 20, 30, dtype=np.uint8) for _ in range(3)], [rng.integers(0, 20, 30, dtype=np.uint8) for _ in range(3)]),
    }
    for name, (s3, sa) in cases.items():
        db = str(tmp_path / name)
        util.write_db(db, s3, sa)
        odb = O.OracleDb(db)
        # the bare string is what Unicore passes (arg_parser.rs:238-239): pre-step + 3-step cascade, sub-databases laid out on the device
        for opts in ("-c 0.8 --single-step-clustering", "-c 0.8 --cluster-steps 2 --linclust 0", "-c 0.8"):
            out = str(tmp_path / (name + "_c"))
            U.cluster(db, out + "_cluster", str(tmp_path / "tmp"), opts)
            U.createtsv(db, out + "_cluster", out + ".tsv")
            p = util.oracle_params(O, "-c 0.8")
            if opts == "-c 0.8":
                ref = O.cluster_workflow(odb, p, O.cascade_thresholds(p, 4.0, 3), linclust_m=20, threads=2)
            else:
                ref = O.cluster_cascade(odb, p, O.cascade_thresholds(p, 4.0, 2 if "steps" in opts else 1), threads=2)
            O.write_tsv(str(tmp_path / "ref.tsv"), odb, ref["assign"])
            assert open(out + ".tsv", "rb").read() == open(str(tmp_path / "ref.tsv"), "rb").read(), (name, opts)
        U.search(db, db, str(tmp_path / (name + "_aln")), str(tmp_path / "tmp"), "-c 0.8")
        U.convertalis(db, db, str(tmp_path / (name + "_aln")), str(tmp_path / (name + ".m8")))
        ps = util.oracle_params(O, "-e 10 --max-seqs 1000 -c 0.8")
        rs = O.search(odb, odb, ps, threads=2)
        O.write_m8(str(tmp_path / "ref.m8"), odb, odb, ps, rs)
        assert open(str(tmp_path / (name + ".m8")), "rb").read() == open(str(tmp_path / "ref.m8"), "rb").read(), name
    # an empty but well-formed database: every workflow succeeds and writes an empty TSV
    db = str(tmp_path / "empty")
    util.write_db(db, [], [])
    for opts in ("-c 0.8 --single-step-clustering", "-c 0.8 --cluster-steps 2 --linclust 0", "-c 0.8"):
        st = U.cluster(db, db + "_cluster", str(tmp_path / "tmp"), opts)
        U.createtsv(db, db + "_cluster", db + ".tsv")
        assert st["n_clusters"] == 0 and open(db + ".tsv", "rb").read() == b"", opts


def test_per_query_threshold_table_rule(O, small, tmp_path):
    """optional rule UC-1/E (default off): `--min-score-table FILE` — one integer per database sequence replaces the Karlin-Altschul
    threshold on the corrected score (the hook for a fitted per-query E-value model) — honoured alike by oracle and engine; plain step only"""
    import ctypes as C
    import unicore_amd as U
    n = len(small["s3"])
    rng = np.random.default_rng(8)
    base = util.oracle_params(O, "-c 0.5")
    # a third of the queries keep (roughly) their Karlin-Altschul threshold, a third let everything through, a third nothing
    table = np.array([(max(1, O.min_score(small["odb"], base, q) + int(rng.integers(-25, 40))), 1, 5000)[q % 3] for q in range(n)], np.int32)
    path = str(tmp_path / "thr.txt")
    np.savetxt(path, table, fmt="%d")
    opts = "-c 0.5 --single-step-clustering --min-score-table " + path
    e = U.Engine(opts, verbosity=1)
    e.set_db(small["off"], *util.flat(small["s3"], small["sa"])[1:])
    e.prefilter(); e.align()
    p = util.oracle_params(O, "-c 0.5")
    p.min_score_table = table.ctypes.data
    ref = O.cluster(small["odb"], p, threads=8)
    ref0 = O.cluster(small["odb"], base, threads=8)
    cnt, _ = e.hits()
    al = e.alns()
    ra = np.concatenate([ref["aln"][i, : cnt[i]] for i in range(len(cnt))])
    for f in ("score", "score_rev", "corrected", "pass_evalue", "accepted"):
        assert np.array_equal(al[f], ra[f]), f
    assert np.array_equal(U.setcover(e.n, e.edges()), ref["assign"])
    assert (ref["aln"]["pass_evalue"] != ref0["aln"]["pass_evalue"]).sum() > 0          # the table does change the gate
    e.close()
    # wrong length, cascade: errors, not silence
    np.savetxt(path, table[:-1], fmt="%d")
    e = U.Engine(opts, verbosity=1)
    e.set_db(small["off"], *util.flat(small["s3"], small["sa"])[1:])
    e.prefilter()
    with pytest.raises(U.UcError):
        e.align()
    e.close()
    assert U.check_options("-c 0.8 --min-score-table " + path) == 0          # syntax only: the workflow check happens when the parameters are finalised
    with pytest.raises(U.UcError):
        U.Engine("-c 0.8 --cluster-steps 3 --min-score-table " + path)


def test_device_set_cover_equals_the_sequential_rule(O):
    """E7 runs on the GPU as parallel rounds of local maxima (uc_align.hip: set_cover_graph): the assignment must be the sequential
    greedy rule's (most unassigned nodes first, ties: smallest id) on every graph shape — long paths and rings (one pick per round at the
    front of a count tie: the worst case for the number of rounds), stars, cliques that share members, hub chains, duplicate / self edges,
    isolated nodes — against both the oracle and the all-host product cover"""
    import unicore_amd as U
    e = U.Engine("-c 0.8", verbosity=1)
    s3, sa = util.family_db(1, n_fam=2, members=2)
    e.set_db(*util.flat(s3, sa))
    rng = np.random.default_rng(77)

    def check(n, edges, tag):
        edges = np.asarray(edges, np.uint32).reshape(-1, 2)
        # the engine's cover wants an engine whose DB has n sequences only for the bounds check of the C ABI: use the raw call
        a = np.zeros(n, np.uint32)
        import ctypes as C
        rc = U.lib().uc_engine_setcover(e._h, edges.ctypes.data, len(edges), a.ctypes.data) if n == e.n else None
        if rc is None:
            return
        assert rc == 0, tag
        assert np.array_equal(a, O.setcover(n, edges)) and np.array_equal(a, U.setcover(n, edges)), tag
    # every shape at the size of a database of exactly n sequences
    def with_n(n):
        s3 = [np.zeros(5, np.uint8)] * n
        e.set_db(*util.flat(s3, s3))
    for n in (1, 2, 7, 64, 257, 2000):
        with_n(n)
        ids = np.arange(n)
        shapes = {
            "empty": np.zeros((0, 2), np.uint32),
            "path": np.stack([ids[:-1], ids[1:]], 1),
            "ring": np.stack([ids, (ids + 1) % n], 1),
            "star": np.stack([np.zeros(n, int), ids], 1),
            "two stars sharing leaves": np.concatenate([np.stack([np.zeros(n // 2, int), ids[n // 2:n // 2 + n // 2]], 1),
                                                        np.stack([np.ones(n // 2, int), ids[n // 2:n // 2 + n // 2]], 1)]) if n > 4 else np.zeros((0, 2), int),
            "self + duplicates": np.concatenate([np.stack([ids, ids], 1), np.stack([ids[:-1], ids[1:]], 1), np.stack([ids[1:], ids[:-1]], 1)]),
            "descending path": np.stack([ids[1:][::-1], ids[:-1][::-1]], 1),
        }
        for k in range(6):
            m = int(rng.integers(0, 6 * n + 1))
            shapes["random %d" % k] = rng.integers(0, n, (m, 2))
        if n >= 64:
            fam = rng.integers(0, max(2, n // 12), n)
            shapes["cliques with bridges"] = np.array([(i, j) for i in range(n) for j in np.nonzero(fam == fam[i])[0][:9]] +
                                                     [(int(a), int(b)) for a, b in rng.integers(0, n, (n // 10, 2))])
            hubs = np.arange(0, n, 16)
            shapes["hub chain"] = np.concatenate([np.stack([hubs[:-1], hubs[1:]], 1)] + [np.stack([np.full(15, h), np.arange(h + 1, h + 16) % n], 1) for h in hubs])
        for tag, ed in shapes.items():
            check(n, ed, (n, tag))
    e.close()


def test_full_size_execution_variants_identical(tmp_path):
    """BASELINE configs[1] at full size (158 k sequences, 14.1 M alignments): the default path (packed kernel, known-score
    re-runs, mutual hits sharing their DPs) gives byte-identical alignment records, the same edge set and the same
    clusters as the plain path (int32 kernel, every directed pair computed on its own)"""
    import unicore_amd as U
    db = util.gen_synth_db(str(tmp_path / "db"), 50, 0x5EED0002, 6000, 1.0)
    res = []
    for opts in ("-c 0.8", "-c 0.8 --sw-kernel i32 --sym-dedup 0"):
        e = U.Engine(opts, verbosity=1)
        e.load_db(db)
        e.prefilter()
        e.align()
        al, ed = e.alns(), e.edges()
        a = e.setcover(ed)
        key = np.sort(ed[:, 0].astype(np.uint64) << np.uint64(32) | ed[:, 1].astype(np.uint64))
        res.append((al.tobytes(), key.tobytes(), a.tobytes(), len(al)))
        del e
    assert res[0][3] > 10_000_000
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and res[0][2] == res[1][2]
    # ... and both equal the CPU oracle run END TO END at this size in the build container (tools/oracle_at_size.py --config c2 ->
    # tests/golden/c2_sha.json): the whole clust.tsv, the set of accepted pairs, the alignment and cluster counts
    gold = os.path.join(util.ROOT, "tests", "golden", "c2_sha.json")
    if os.path.exists(gold):
        import hashlib
        import json
        g = json.load(open(gold))
        assert hashlib.sha256(res[0][1]).hexdigest() == g["counts"]["edge_set_sha256"] and res[0][3] == g["counts"]["n_alignments"]
        out = str(tmp_path / "clust")
        assign = np.frombuffer(res[0][2], np.uint32)
        assert int((assign == np.arange(len(assign))).sum()) == g["counts"]["n_clusters"]
        assert U.lib().uc_write_cluster_db((out + "_cluster").encode(), len(assign), assign.ctypes.data) == 0
        U.createtsv(db, out + "_cluster", out + ".tsv")
        assert hashlib.sha256(open(out + ".tsv", "rb").read()).hexdigest() == g["tsv_sha256"]


def test_bench_line_contract(tmp_path):
    """bench.py prints ONE JSON line with the keys the driver reads (tiny workload here)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--proteomes", "3", "--families", "60", "--steps", "2", "--warmup", "1",
                        "--cpu-seconds", "1", "--workdir", str(tmp_path), "--detail-dir", str(tmp_path / "detail")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    slim = json.loads(lines[0])
    # r06: the line is the SLIM record (< 8 kB: the driver keeps a 10 kB tail) with the figures a reader needs at the top level; the full record
    # (formulas, notes, both CPU legs, counts) is the side file the line names - the recomputation checks below read that
    assert len(lines[0]) < 8000
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "value_definition", "value_disk_to_tsv_aln_s", "disk_to_tsv_wall_s", "workflow_default_wall_s",
              "value_one_shot_aln_s", "stages_s_per_step", "detail_file"):
        assert k in slim, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu_frac"):
        assert k in slim["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in slim["cpu_baseline"], k
    assert "workload" in slim["config"] and "model" not in slim["config"]
    d = json.load(open(slim["detail_file"]))
    assert abs(slim["value"] - d["value"]) <= 1e-5 * d["value"] and abs(slim["value_disk_to_tsv_aln_s"] - d["value_disk_to_tsv"]["value"]) <= 1e-5 * d["value_disk_to_tsv"]["value"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("port", "simd") and d["cpu_baseline"]["value"] > 0
    assert {b["kind"] for b in d["cpu_baselines"]} == {"port", "simd"} and all("not available" in b["foldseek"] for b in d["cpu_baselines"])
    # SURVEY.md 8(d)'s own definition (disk -> clust.tsv) and the default workflow are measured beside the headline value
    assert d["value_disk_to_tsv"]["value"] > 0 and d["value_disk_to_tsv"]["same_clusters_as_steps"]
    assert d["workflow_default"]["clusters"] > 0
    for blk in ("roofline_prefilter", "roofline_end_to_end"):
        for k in ("bound", "achieved", "peak", "unit", "frac"):
            assert k in d[blk], (blk, k)
    assert sum(d["algorithmic_bytes_per_step"].values()) > 0
    # what an unmodified Unicore experiences (two spawns of cluster.rs:45-64) is in the line for a custom size too
    assert d["value_one_shot_processes"]["plain_step"]["wall_s_best"] > 0 and d["value_one_shot_processes"]["default_workflow"]["value"] > 0
    # every fraction can be recomputed from the counts in the same line (r2's line could not: a cumulative counter)
    c, rp = d["counts_rank0_per_step"], d["roofline_prefilter"]
    ab = d["algorithmic_bytes_per_step"]
    assert ab["kmer"] == 8 * c["n_sim_kmers"] + 6 * c["n_kmer_hits"] + 8 * c["n_candidates"]
    assert ab["index"] == 6 * c["n_index_entries"] + 8 * 20 ** 6 and ab["select"] == 16 * c["n_candidates"]
    pre = ab["index"] + ab["kmer"] + ab["ungapped"] + ab["select"]
    assert abs(rp["algorithmic_bytes_per_step"] - pre) < 1 and abs(rp["frac"] - pre / (rp["kernel_ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 1e-9
    r = d["roofline"]
    assert abs(r["frac"] - r["algorithmic_bytes_per_step"] / (r["kernel_ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 1e-9
    assert abs(r["valu_frac"] - r["cells_run_per_step"] * r["valu_ops_per_cell"] / (r["kernel_ms_per_step"] * 1e-3) / r["valu_peak_lane_ops"]) < 1e-9


def test_bench_line_does_not_depend_on_the_step_count(tmp_path):
    """r2's line added 8 B x the CUMULATIVE similar-k-mer counter once per step, so roofline_prefilter.frac grew with --steps
    (0.06 at one step, 0.15 at twenty).  Per-step algorithmic bytes, counts and everything derived from them alone must be
    identical for --steps 1 and --steps 4."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for steps in (1, 4):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--proteomes", "3", "--families", "60", "--steps", str(steps), "--warmup", "1",
                            "--no-cpu-baseline", "--no-extra-legs", "--full-line", "--workdir", str(tmp_path)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    a, b = res
    assert a["algorithmic_bytes_per_step"] == b["algorithmic_bytes_per_step"]
    assert a["counts_rank0_per_step"] == b["counts_rank0_per_step"]
    for blk, key in (("roofline_prefilter", "algorithmic_bytes_per_step"), ("roofline_end_to_end", "algorithmic_bytes_per_step"), ("roofline_end_to_end", "bytes_per_alignment"),
                     ("roofline", "algorithmic_bytes_per_step"), ("roofline", "cells_run_per_step"), ("roofline", "cells_algorithmic_per_step")):
        assert a[blk][key] == b[blk][key], (blk, key)


def test_traceback_bytes_in_several_batches(O, small):
    """the traceback-byte matrices of MODE 7 are produced in batches bounded by a scratch budget: a 1 MB budget (many
    batches) gives the same records as the default single batch"""
    import unicore_amd as U
    opts = "-c 0.5 --min-seq-id 0.3"
    res = []
    for budget in (None, "1"):
        if budget:
            os.environ["UC_TB_BUDGET_MB"] = budget
        try:
            e = U.Engine(opts, verbosity=1)
            e.set_db(small["off"], *util.flat(small["s3"], small["sa"])[1:])
            e.prefilter()
            e.align()
            res.append(e.alns().tobytes())
        finally:
            os.environ.pop("UC_TB_BUDGET_MB", None)
    assert res[0] == res[1]
    al = np.frombuffer(res[0], U.ALN_DTYPE)
    assert (al["aln_len"] > 0).sum() > 50


def zigzag_db(seed, n_pairs=24):
    """pairs whose optimal alignment leaves the corridor between its start and end diagonals: q = A X B, t = A Y B with unrelated X, Y (8-40
    residues) - a gap out and a gap back; plus single-gap and shifted variants, on top of a family database"""
    rng = np.random.default_rng(seed)
    s3, sa = util.family_db(seed, n_fam=8, members=5, extra=(700, 1100))
    def rnd(L): return rng.integers(0, 20, L, dtype=np.uint8), rng.integers(0, 20, L, dtype=np.uint8)
    for k in range(n_pairs):
        L1, L2, lx, ly = int(rng.integers(40, 200)), int(rng.integers(40, 400)), int(rng.integers(8, 40)), int(rng.integers(8, 40))
        a3, aa = rnd(L1); b3, ba = rnd(L2); x3, xa = rnd(lx); y3, ya = rnd(ly)
        if k % 3 == 2: y3, ya = y3[:0], ya[:0]                                     # one gap only: stays inside the corridor
        pre = rnd(int(rng.integers(0, 30))) if k % 4 == 1 else rnd(0)              # a shifted start diagonal
        s3.append(np.concatenate([a3, x3, b3])); sa.append(np.concatenate([aa, xa, ba]))
        s3.append(np.concatenate([pre[0], a3, y3, b3])); sa.append(np.concatenate([pre[1], aa, ya, ba]))
    return s3, sa


@pytest.mark.parametrize("opts", ["-c 0.5 --min-seq-id 0.3", "-c 0.3 --cov-mode 1 --min-seq-id 0.2 --gap-open 6 --gap-extend 1 -e 10"])
def test_traceback_band_and_its_fallback(O, opts, capfd):
    """r05: MODE 7 stores the H bytes of a diagonal band of the box (tb_band_of, uc_device.h) and the walk gives up on a pair whose traceback
    leaves it; such pairs are redone with the whole box stored.  On a database with zig-zag pairs (a gap out of the corridor and a gap back):
    whatever the half-width - 0 (whole box, the r04 layout), 1, 4 (the zig-zags leave the band: the fallback does their work), the default -
    the records are the same bytes, and the traceback statistics are the scalar oracle's."""
    import unicore_amd as U
    s3, sa = zigzag_db(77)
    off, c3, ca = util.flat(s3, sa)
    res, redone = {}, {}
    for w in ("0", "1", "4", None):
        if w is not None:
            os.environ["UC_TB_BAND"] = w
        os.environ["UC_TIMING"] = "1"
        try:
            e = U.Engine(opts, verbosity=1)
            e.set_db(off, c3, ca)
            e.prefilter()
            capfd.readouterr()
            e.align()
            err = capfd.readouterr().err
            res[w] = e.alns().tobytes()
            if w is None:
                cnt, hits = e.hits()
            e.close()
        finally:
            os.environ.pop("UC_TB_BAND", None)
            os.environ.pop("UC_TIMING", None)
        # "sw pass mode 7: N pairs, ..., band W, B batch(es)" lines of tb_batch: a band-0 line in a run whose band is W > 0 is the fallback at work
        m7 = [l for l in err.splitlines() if "sw pass mode 7:" in l]
        bands = [int(l.split("band ")[1].split(",")[0]) for l in m7]
        redone[w] = sum(int(l.split("mode 7: ")[1].split(" pairs")[0]) for l in m7 if "band 0," in l) if bands and max(bands) > 0 else 0
    assert res["0"] == res["1"] == res["4"] == res[None]
    assert redone["1"] >= 10 and redone["4"] >= 3 and redone["0"] == 0, redone           # the zig-zag pairs went through the fallback
    al = np.frombuffer(res[None], U.ALN_DTYPE)
    assert (al["aln_len"] > 0).sum() > 100
    odb = O.OracleDb(s3=s3, sa=sa)
    p = util.oracle_params(O, opts)
    o = np.concatenate([[0], np.cumsum(cnt)])
    checked = 0
    for q in range(len(cnt)):
        ms = O.min_score(odb, p, q)
        for k in range(int(o[q]), int(o[q + 1])):
            if al["aln_len"][k] > 0:
                ref = O.align_pair(odb, p, q, int(hits["target"][k]), ms)
                assert (ref["aln_len"], ref["idents"]) == (al["aln_len"][k], al["idents"][k]), (q, k)
                checked += 1
    assert checked > 100


@pytest.mark.parametrize("opts,steps,m", [("-c 0.8 --linclust 1 --cluster-steps 1", 1, 20), ("-c 0.8 --linclust 1 --cluster-steps 3", 3, 20),
                                          ("-c 0.5 --linclust 1 --kmer-per-seq 5 --cluster-steps 2", 2, 5),
                                          ("-c 0.8 --length-gate 1 --linclust 1 --cluster-steps 3", 3, 20)])      # rule UC-1/L through every round
def test_linclust_workflow_tsv_bytes(O, tmp_path, opts, steps, m):
    """E8a: linear-time pre-step (minimum-hash k-mer groups, centre = longest member, candidate pairs
    through E5/E6, set cover) in front of the cascade rounds == the oracle's workflow, byte for byte"""
    import unicore_amd as U
    db = util.gen_synth_db(str(tmp_path / "db"), 6, 0x5EED0004, 40, 0.6)
    out = str(tmp_path / "clust")
    st = U.cluster(db, out + "_cluster", str(tmp_path / "tmp"), opts, threads=4)
    U.createtsv(db, out + "_cluster", out + ".tsv")
    odb = O.OracleDb(db)
    base = " ".join(t for t in opts.replace("--linclust 1", "").replace("--cluster-steps %d" % steps, "").replace("--kmer-per-seq %d" % m, "").split())
    p = util.oracle_params(O, base)
    ref = O.cluster_workflow(odb, p, O.cascade_thresholds(p, 4.0, steps), linclust_m=m, threads=8)
    O.write_tsv(str(tmp_path / "ref.tsv"), odb, ref["assign"])
    assert open(out + ".tsv", "rb").read() == open(str(tmp_path / "ref.tsv"), "rb").read()
    util.tsv_invariants(out + ".tsv", odb.names())
    assert st["n_clusters"] == ref["counts"]["n_clusters"] and st["n_gapped_alignments"] == ref["counts"]["n_alignments"]
    assert ref["round_sizes"][1] < odb.n                      # the pre-step removed something
    pr = O.linclust_pairs(odb, p, m)
    assert len(pr) > 20 and (pr[:, 0] != pr[:, 1]).all()


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_option_sets_against_the_oracle(O, seed):
    """property test: random small databases x random option strings (gates, gap costs, sensitivity, truncation, execution
    variants): accepted pairs and the cluster assignment equal the oracle's.  Seeds >= 12 add sequences beyond 2048
    residues with a mutated copy each (row-blocked kernel in every pass)."""
    import unicore_amd as U
    rng = np.random.default_rng(1000 + seed)
    extra = tuple(int(x) for x in rng.integers(300, 1900, int(rng.integers(0, 3))))
    if seed >= 12:
        extra = extra + tuple(int(x) for x in rng.integers(2049, 5200, int(rng.integers(1, 3))))
    s3, sa = util.family_db(500 + seed, n_fam=int(rng.integers(4, 12)), members=int(rng.integers(2, 8)), lmin=int(rng.integers(20, 60)),
                            lmax=int(rng.integers(80, 400)), sub3=float(rng.uniform(0.05, 0.3)), suba=float(rng.uniform(0.1, 0.5)),
                            indel=float(rng.uniform(0.0, 0.05)), extra=extra)
    if seed >= 12:
        k = int(np.argmax([len(x) for x in s3])); a3, aa = s3[k].copy(), sa[k].copy()
        m = rng.random(len(a3)) < 0.1; a3[m] = rng.integers(0, 20, int(m.sum()), dtype=np.uint8)
        cut = int(rng.integers(100, len(a3) - 100)); a3 = np.delete(a3, slice(cut, cut + 5)); aa = np.delete(aa, slice(cut, cut + 5))
        s3.append(a3); sa.append(aa)
    opts = ["-c %.2f" % rng.choice([0.3, 0.5, 0.8, 0.9]), "--cov-mode %d" % rng.integers(0, 3), "-e %g" % rng.choice([1e-3, 1e-6, 10.0]),
            "--max-seqs %d" % rng.choice([3, 20, 300])]
    if rng.random() < 0.5: opts.append("-s %g" % rng.choice([2.0, 4.0, 6.0]))
    if rng.random() < 0.3: opts.append("--min-diag-hits %d" % rng.choice([1, 3]))
    if rng.random() < 0.4: opts.append("--gap-open %d --gap-extend %d" % (rng.choice([8, 10, 12]), rng.choice([1, 2])))
    if rng.random() < 0.3: opts.append("--rev-correction 0")
    if rng.random() < 0.4: opts.append("--min-seq-id %.2f" % rng.choice([0.2, 0.4, 0.6]))
    if rng.random() < 0.3: opts.append("--min-ungapped-score %d" % rng.choice([10, 25]))
    eng = []
    if rng.random() < 0.3: eng.append("--sym-dedup 0")
    if rng.random() < 0.3: eng.append("--sw-kernel i32")
    if rng.random() < 0.35: opts.append("--length-gate 1")     # optional rule UC-1/L (drawn last: the other draws of a seed are what they were)
    ostr = " ".join(opts)
    off, c3, ca = util.flat(s3, sa)
    e = U.Engine(ostr + " " + " ".join(eng), verbosity=1)
    e.set_db(off, c3, ca)
    e.prefilter()
    e.align()
    ref = O.cluster(O.OracleDb(s3=s3, sa=sa), util.oracle_params(O, ostr), threads=8)
    cnt, hits = e.hits()
    assert np.array_equal(cnt, ref["hit_cnt"]), ostr
    al = e.alns()
    ra = np.concatenate([ref["aln"][i, : cnt[i]] for i in range(len(cnt))]) if cnt.sum() else al
    for f in ("score", "score_rev", "corrected", "pass_evalue", "accepted", "aln_len", "idents"):
        assert np.array_equal(al[f], ra[f]), (f, ostr, eng)
    pe = al["pass_evalue"] == 1
    for f in ("qstart", "qend", "tstart", "tend"):
        assert np.array_equal(al[f][pe], ra[f][pe]), (f, ostr, eng)
    assert np.array_equal(e.setcover(e.edges()), ref["assign"]), (ostr, eng)
