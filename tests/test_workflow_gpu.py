"""The DEFAULT workflow — what `unicore cluster` actually triggers — as a first-class, at-size parity case.

cluster.rs:35,45-49 forwards the user's option string (default "-c 0.8", arg_parser.rs:238-239) WITHOUT --single-step-clustering, so the
`foldseek cluster` call runs a linear-time pre-step and a 3-step cascade, every round on the representatives of the round before and with
the sensitivity rising from 1 to the target.  These tests run the whole call (uc_cluster from the DB files, one C entry point) and use the
workflow observer of the C ABI (uc_set_round_hook) to look INTO every round: the round's sequence set, its k-mer threshold, and the hit lists
and alignment records of a random query sample, which are then recomputed by the CPU oracle on that round's sub-database.

  small   every query of every round against the oracle + clust.tsv bytes against uco_cluster_workflow
  c2      BASELINE configs[1]: 50 proteomes, "-c 0.8"
  c3      BASELINE configs[2]: 500 proteomes, "-c 0.8"
  c4-200  BASELINE configs[3]'s options "-c 0.8 --min-seq-id 0.3 -s 7.5" on 200 proteomes
  c4-500  ... on 500 proteomes (1.59 M sequences)
  c4      ... at the NOMINAL 2000 proteomes (6.3 M sequences, 1.9 G residues) — the configuration VERDICT r3 listed as never run"""
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle_py
    return oracle_py


def run_with_round_samples(U, db, out, tmp, opts, per_round, seed, threads=16):
    """uc_cluster(db, opts) with the observer registered -> (stats, [round records]); a record holds the round's ids, threshold, pair count and
    (query, hits, alignment records) of `per_round` random round-local queries (all of them if per_round is None)"""
    recs, errors = [], []

    def hook(rnd, ids, kthr, view):
        try:
            m = len(ids)
            rng = np.random.default_rng(seed + 7 * (rnd + 1))
            qs = np.arange(m) if per_round is None or per_round >= m else np.sort(rng.choice(m, per_round, replace=False))
            got = []
            for q in qs:
                cnt, hits = view.hits_range(int(q), int(q) + 1)
                al = view.alns_range(int(q), int(q) + 1)
                assert int(cnt[0]) == len(hits) == len(al)
                got.append((int(q), hits.copy(), al.copy()))
            recs.append(dict(round=rnd, ids=ids, kmer_thr=kthr, n_pairs=view.hits_size(), samples=got))
        except Exception as e:          # an exception cannot cross the C frame: keep it for the caller
            errors.append(e)

    U.set_round_hook(hook)
    try:
        st = U.cluster(db, out, tmp, opts, threads=threads)
    finally:
        U.set_round_hook(None)
    if errors:
        raise errors[0]
    return st, recs


def check_rounds(O, odb, base_opts, recs, steps=3, m=20, target_s=4.0):
    """every sampled query of every round == the oracle on that round's sub-database; returns the number of records compared"""
    p0 = util.oracle_params(O, base_opts)
    thr = O.cascade_thresholds(p0, target_s, steps)
    assert [r["round"] for r in recs] == [-1] + list(range(steps))
    assert np.array_equal(recs[0]["ids"], np.arange(odb.n))
    compared = 0
    prev = None
    for r in recs:
        ids = r["ids"]
        if prev is not None:
            assert len(ids) <= len(prev) and np.all(np.diff(ids.astype(np.int64)) > 0) and np.isin(ids, prev).all()   # representatives of the round before
        prev = ids
        sub = odb.subset(ids)
        p = util.oracle_params(O, base_opts)
        if r["round"] >= 0:
            assert r["kmer_thr"] == thr[r["round"]], (r["round"], r["kmer_thr"], thr)     # the sensitivity schedule, restated by the test
            p.kmer_thr = thr[r["round"]]
        qs = np.array([s[0] for s in r["samples"]], np.uint32)
        if r["round"] < 0:
            pairs = O.linclust_pairs(sub, p, m)
            assert r["n_pairs"] == len(pairs)
            lo = np.searchsorted(pairs[:, 0], qs, "left"); hi = np.searchsorted(pairs[:, 0], qs, "right")
            for k, (q, hits, al) in enumerate(r["samples"]):
                exp = pairs[lo[k]:hi[k], 1]
                assert np.array_equal(hits["target"], exp), ("pre-step members", q)
                ms = O.min_score(sub, p, q)
                for h in range(len(exp)):
                    ref = O.align_pair(sub, p, q, int(exp[h]), ms)
                    for f in ("score", "score_rev", "corrected", "pass_evalue", "accepted"):
                        assert al[f][h] == ref[f], ("pre-step", q, int(exp[h]), f)
                    compared += 1
        else:
            ix = O.build_index(sub, p)
            _, _, _, ocnt, ohits, oalns = O.simd_sample_run(sub, ix, p, qs, threads=0, records=True)
            O.free_index(ix)
            for k, (q, hits, al) in enumerate(r["samples"]):
                c = int(ocnt[k])
                assert len(hits) == c, ("round", r["round"], q, len(hits), c)
                assert np.array_equal(hits["target"], ohits[k, :c]["t"]) and np.array_equal(hits["score"], ohits[k, :c]["score"]) \
                    and np.array_equal(hits["diag"], ohits[k, :c]["diag"]), ("round", r["round"], q)
                ref = oalns[k, :c]
                for f in ("score", "score_rev", "corrected", "pass_evalue", "accepted"):
                    assert np.array_equal(al[f], ref[f]), ("round", r["round"], q, f)
                pe = ref["pass_evalue"] == 1
                for f in ("qstart", "qend", "tstart", "tend"):
                    assert np.array_equal(al[f][pe], ref[f][pe]), ("round", r["round"], q, f)
                if p.min_seq_id > 0:
                    cov = ref["aln_len"] > 0
                    for f in ("aln_len", "idents"):
                        assert np.array_equal(al[f][cov], ref[f][cov]), ("round", r["round"], q, f)
                compared += c
        del sub
    return compared


def test_round_hook_every_round_equals_the_oracle(O, tmp_path):
    """small database: EVERY query of EVERY round (pre-step + 3 cascade rounds) against the oracle on the round's sub-database, the final
    clust.tsv against uco_cluster_workflow byte for byte, and the observer unregisters cleanly"""
    import unicore_amd as U
    db = util.gen_synth_db(str(tmp_path / "db"), 6, 0x5EED0004, 40, 0.6)
    out = str(tmp_path / "clust")
    for opts, s in (("-c 0.8", 4.0), ("-c 0.8 --min-seq-id 0.3 -s 7.5", 7.5)):
        st, recs = run_with_round_samples(U, db, out + "_cluster", str(tmp_path / "tmp"), opts, None, 1, threads=4)
        U.createtsv(db, out + "_cluster", out + ".tsv")
        odb = O.OracleDb(db)
        assert check_rounds(O, odb, opts, recs, target_s=s) > 300
        p = util.oracle_params(O, opts)
        ref = O.cluster_workflow(odb, p, O.cascade_thresholds(p, s, 3), linclust_m=20, threads=8)
        O.write_tsv(str(tmp_path / "ref.tsv"), odb, ref["assign"])
        assert open(out + ".tsv", "rb").read() == open(str(tmp_path / "ref.tsv"), "rb").read()
        assert [len(r["ids"]) for r in recs] == [int(x) for x in ref["round_sizes"]]
        assert st["n_gapped_alignments"] == ref["counts"]["n_alignments"] == sum(r["n_pairs"] for r in recs)
    # unregistered: a further call sees no hook
    seen = []
    U.set_round_hook(lambda *a: seen.append(a))
    U.set_round_hook(None)
    U.cluster(db, out + "_cluster", str(tmp_path / "tmp"), "-c 0.8", threads=4)
    assert not seen
    # the plain step reports itself as round 0 of a one-round workflow
    st, recs = run_with_round_samples(U, db, out + "_cluster", str(tmp_path / "tmp"), "-c 0.8 --single-step-clustering", 5, 3, threads=4)
    assert [r["round"] for r in recs] == [0] and len(recs[0]["ids"]) == st["n_seqs"]


AT_SIZE = {
    "c2": dict(proteomes=50, seed=0x5EED0002, opts="-c 0.8", s=4.0, per_round=300, min_aln=2_000_000),
    "c3": dict(proteomes=500, seed=0x5EED0003, opts="-c 0.8", s=4.0, per_round=500, min_aln=50_000_000),
    "c4-200": dict(proteomes=200, seed=0x5EED0004, opts="-c 0.8 --min-seq-id 0.3 -s 7.5", s=7.5, per_round=500, min_aln=20_000_000),
    "c4-500": dict(proteomes=500, seed=0x5EED0004, opts="-c 0.8 --min-seq-id 0.3 -s 7.5", s=7.5, per_round=500, min_aln=100_000_000),
    # BASELINE configs[3] at its NOMINAL size.  One pass takes minutes; the oracle side (four sub-database indexes over up to 1.9 G residues and
    # 4 x 300 sampled queries) a few more.
    "c4": dict(proteomes=2000, seed=0x5EED0004, opts="-c 0.8 --min-seq-id 0.3 -s 7.5", s=7.5, per_round=300, min_aln=400_000_000),
    # optional rule UC-1/L (length gate, default off) through every round at configs[2] size, against tests/golden/c3-gate_workflow_sha.json
    # (the CPU oracle's workflow with the rule on, end to end); behind UC_TEST_AT_SIZE_EXTRA=1 - builder-run, log under profiles/
    "c3-gate": dict(proteomes=500, seed=0x5EED0003, opts="-c 0.8 --length-gate 1", s=4.0, per_round=500, min_aln=20_000_000),
}


@pytest.mark.parametrize("name", list(AT_SIZE))
def test_default_workflow_at_size(name, O, tmp_path_factory):
    """the call of cluster.rs:45-49 at BASELINE's sizes: uc_cluster(db, "<options>") with no single-step flag -> pre-step + 3-step cascade;
    (a) a random query sample of EACH round equals the CPU oracle on that round's sub-database (hit lists, scores, gates, coordinates,
    traceback statistics); (b) the TSV satisfies the consumer contract of profile.rs; (c) the rounds shrink and the counters add up."""
    import unicore_amd as U
    cfg = AT_SIZE[name]
    if name == "c4" and os.environ.get("UC_TEST_NOMINAL_C4") != "1":
        pytest.skip("the nominal 2000-proteome case takes ~15 min (5 min on the GPU, the rest in the CPU oracle's four sub-database passes): "
                    "set UC_TEST_NOMINAL_C4=1; the builder's run is committed as profiles/r04/test_workflow_c4_nominal.log")
    if name == "c3-gate" and not os.environ.get("UC_TEST_AT_SIZE_EXTRA"):
        pytest.skip("set UC_TEST_AT_SIZE_EXTRA=1 (the builder's run: profiles/r04/gpu_test_workflow_c3_gate.log)")
    d = tmp_path_factory.mktemp("wf_" + name.replace("-", "_"))
    db = util.gen_synth_db(str(d / "db"), cfg["proteomes"], cfg["seed"], 6000, 1.0)
    out = str(d / "clust")
    st, recs = run_with_round_samples(U, db, out + "_cluster", str(d / "tmp"), cfg["opts"], cfg["per_round"], 20260929)
    n = st["n_seqs"]
    listed = sum(r["n_pairs"] for r in recs)      # under the length gate (UC-1/L) the listed pairs that are ruled out are no alignments
    assert st["n_gapped_alignments"] >= cfg["min_aln"] and (st["n_gapped_alignments"] < listed if "--length-gate 1" in cfg["opts"] else st["n_gapped_alignments"] == listed)
    sizes = [len(r["ids"]) for r in recs]
    assert sizes[0] == n and all(a >= b for a, b in zip(sizes, sizes[1:])) and sizes[1] < n
    assert 0 < st["n_clusters"] <= sizes[-1]
    U.createtsv(db, out + "_cluster", out + ".tsv")
    U.lib().uc_release_scratch()
    names = [l.split("\t")[1] for l in open(db + ".lookup")]
    rows = util.tsv_invariants(out + ".tsv", names)
    assert len(rows) == n and len({r[0] for r in rows}) == st["n_clusters"]
    del rows, names
    # the WHOLE clust.tsv, the round sizes and the summed stage counters against the CPU oracle's workflow run end to end at this size in the build
    # container (tools/oracle_at_size.py --workflow -> tests/golden/<name>_workflow_sha.json), where that run exists
    gold = os.path.join(util.ROOT, "tests", "golden", "%s_workflow_sha.json" % name)
    if os.path.exists(gold):
        import hashlib
        import json
        g = json.load(open(gold))
        data = open(out + ".tsv", "rb").read()
        assert g["sequences"] == n and g["round_sizes"] == sizes
        assert len(data) == g["tsv_bytes"] and hashlib.sha256(data).hexdigest() == g["tsv_sha256"], "workflow clust.tsv differs from the CPU oracle's at full size"
        for a, b in (("n_sim_kmers", "n_sim_kmers"), ("n_kmer_hits", "n_kmer_hits"), ("n_candidates", "n_candidates"), ("n_gapped_alignments", "n_alignments"),
                     ("cells_fwd", "cells_fwd"), ("cells_rev", "cells_rev"), ("cells_start", "cells_start"), ("n_clusters", "n_clusters")):
            assert st[a] == g["counts"][b], (a, st[a], g["counts"][b])
    odb = O.OracleDb(db)
    assert odb.n == n
    compared = check_rounds(O, odb, cfg["opts"], recs, target_s=cfg["s"])
    assert compared > 20 * cfg["per_round"]
