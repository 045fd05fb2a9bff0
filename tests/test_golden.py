"""Committed oracle-generated fixtures (tests/golden/, generator make_golden.py alongside).
CPU: the oracle still reproduces them (pins the oracle against regressions) and the TSV satisfies the
consumer contract of src/modules/profile.rs.  GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

import util

GOLD = os.path.join(util.ROOT, "tests", "golden")
CASES = ["default", "sensitive"]


def load(name):
    z = np.load(os.path.join(GOLD, "expected_%s.npz" % name))
    return z, str(z["options"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name, tmp_path):
    from oracle import oracle_py as O
    z, opts = load(name)
    odb = O.OracleDb(os.path.join(GOLD, "db"))
    r = O.cluster(odb, util.oracle_params(O, opts), threads=4)
    cnt = r["hit_cnt"]
    assert np.array_equal(cnt, z["hit_cnt"]) and np.array_equal(r["assign"], z["assign"])
    hits = np.concatenate([r["hits"][q, : cnt[q]] for q in range(odb.n)])
    aln = np.concatenate([r["aln"][q, : cnt[q]] for q in range(odb.n)])
    assert hits.tobytes() == z["hits"].tobytes() and aln.tobytes() == z["aln"].tobytes()
    O.write_tsv(str(tmp_path / "c.tsv"), odb, r["assign"])
    assert open(tmp_path / "c.tsv", "rb").read() == open(os.path.join(GOLD, "clust_%s.tsv" % name), "rb").read()


@pytest.mark.parametrize("name", CASES)
def test_golden_tsv_satisfies_the_consumer_contract(name):
    names = [l.split("\t")[1] for l in open(os.path.join(GOLD, "db.lookup"))]
    rows = util.tsv_invariants(os.path.join(GOLD, "clust_%s.tsv" % name), names)
    mapped = {l.split("\t")[0] for l in open(os.path.join(GOLD, "db.map"))}
    assert {r[1] for r in rows} <= mapped          # names equal col 0 of <db>.map (profile.rs:22,79)
    assert all(n.startswith("unicore_") and len(n) == 18 for n in names)   # createdb.rs:104-106


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_path_reproduces_golden(name, tmp_path):
    import unicore_amd as U
    z, opts = load(name)
    e = U.Engine(opts, verbosity=1)
    e.load_db(os.path.join(GOLD, "db"))
    e.prefilter()
    cnt, hits = e.hits()
    assert np.array_equal(cnt, z["hit_cnt"])
    assert np.array_equal(hits["target"], z["hits"]["t"]) and np.array_equal(hits["score"], z["hits"]["score"]) and np.array_equal(hits["diag"], z["hits"]["diag"])
    e.align()
    al = e.alns()
    for f in ("score", "score_rev", "corrected", "pass_evalue", "accepted"):
        assert np.array_equal(al[f], z["aln"][f]), f
    pe = al["pass_evalue"] == 1
    for f in ("qstart", "qend", "tstart", "tend"):
        assert np.array_equal(al[f][pe], z["aln"][f][pe]), f
    assert np.array_equal(U.setcover(e.n, e.edges()), z["assign"])
    out = str(tmp_path / "clust")
    U.cluster(os.path.join(GOLD, "db"), out + "_cluster", str(tmp_path / "tmp"), opts + " --single-step-clustering")
    U.createtsv(os.path.join(GOLD, "db"), out + "_cluster", out + ".tsv")
    assert open(out + ".tsv", "rb").read() == open(os.path.join(GOLD, "clust_%s.tsv" % name), "rb").read()


def test_oracle_reproduces_golden_cascade_and_search(tmp_path):
    from oracle import oracle_py as O
    odb = O.OracleDb(os.path.join(GOLD, "db"))
    p = util.oracle_params(O, "-c 0.8")
    rc = O.cluster_cascade(odb, p, O.cascade_thresholds(p, 4.0, 3), threads=4)
    O.write_tsv(str(tmp_path / "c.tsv"), odb, rc["assign"])
    assert open(tmp_path / "c.tsv", "rb").read() == open(os.path.join(GOLD, "clust_cascade3.tsv"), "rb").read()
    rw = O.cluster_workflow(odb, p, O.cascade_thresholds(p, 4.0, 3), linclust_m=20, threads=4)
    O.write_tsv(str(tmp_path / "w.tsv"), odb, rw["assign"])
    assert open(tmp_path / "w.tsv", "rb").read() == open(os.path.join(GOLD, "clust_linclust_cascade3.tsv"), "rb").read()
    ps = util.oracle_params(O, "-e 10 --max-seqs 1000 -c 0.8")
    rs = O.search(odb, odb, ps, threads=4)
    O.write_m8(str(tmp_path / "s.m8"), odb, odb, ps, rs)
    assert open(tmp_path / "s.m8", "rb").read() == open(os.path.join(GOLD, "search_self.m8"), "rb").read()


@pytest.mark.gpu
def test_hip_path_reproduces_golden_cascade_and_search(tmp_path):
    import unicore_amd as U
    db = os.path.join(GOLD, "db")
    U.cluster(db, str(tmp_path / "c_cluster"), str(tmp_path / "tmp"), "-c 0.8 --cluster-steps 3 --linclust 0")
    U.createtsv(db, str(tmp_path / "c_cluster"), str(tmp_path / "c.tsv"))
    assert open(tmp_path / "c.tsv", "rb").read() == open(os.path.join(GOLD, "clust_cascade3.tsv"), "rb").read()
    U.cluster(db, str(tmp_path / "w_cluster"), str(tmp_path / "tmp"), "-c 0.8 --linclust 1 --cluster-steps 3")
    U.createtsv(db, str(tmp_path / "w_cluster"), str(tmp_path / "w.tsv"))
    assert open(tmp_path / "w.tsv", "rb").read() == open(os.path.join(GOLD, "clust_linclust_cascade3.tsv"), "rb").read()
    # a bare "-c 0.8" - all that Unicore ever passes (arg_parser.rs:238-239) - runs that same default workflow
    U.cluster(db, str(tmp_path / "d_cluster"), str(tmp_path / "tmp"), "-c 0.8")
    U.createtsv(db, str(tmp_path / "d_cluster"), str(tmp_path / "d.tsv"))
    assert open(tmp_path / "d.tsv", "rb").read() == open(os.path.join(GOLD, "clust_linclust_cascade3.tsv"), "rb").read()
    U.search(db, db, str(tmp_path / "s_aln"), str(tmp_path / "tmp"), "-c 0.8")
    U.convertalis(db, db, str(tmp_path / "s_aln"), str(tmp_path / "s.m8"))
    assert open(tmp_path / "s.m8", "rb").read() == open(os.path.join(GOLD, "search_self.m8"), "rb").read()


C1 = os.path.join(GOLD, "c1")


def test_c1_example_data_oracle_and_consumer_contract(tmp_path):
    """BASELINE configs[0] as a parity case: 5 proteomes of the reference's example/data in createdb format (stand-in 3Di
    track, see make_c1.py): the oracle reproduces the committed clust.tsv, which satisfies profile.rs' requirements"""
    from oracle import oracle_py as O
    odb = O.OracleDb(os.path.join(C1, "db"))
    r = O.cluster(odb, util.oracle_params(O, "-c 0.8"), threads=4, dumps=False)
    O.write_tsv(str(tmp_path / "c.tsv"), odb, r["assign"])
    assert open(tmp_path / "c.tsv", "rb").read() == open(os.path.join(C1, "clust.tsv"), "rb").read()
    names = [l.split("\t")[1] for l in open(os.path.join(C1, "db.lookup"))]
    rows = util.tsv_invariants(os.path.join(C1, "clust.tsv"), names)
    mapped = {l.split("\t")[0] for l in open(os.path.join(C1, "db.map"))}
    species = {l.split("\t")[1] for l in open(os.path.join(C1, "db.map"))}
    assert {r_[1] for r_ in rows} <= mapped and len(species) == 5
    assert len({r_[0] for r_ in rows}) < len(names)          # orthologs of different species were clustered
    # the default workflow (what a bare "-c 0.8" runs: pre-step + 3-step cascade) has its own committed result
    rw = O.cluster_workflow(odb, util.oracle_params(O, "-c 0.8"), O.cascade_thresholds(util.oracle_params(O, "-c 0.8"), 4.0, 3), linclust_m=20, threads=4)
    O.write_tsv(str(tmp_path / "w.tsv"), odb, rw["assign"])
    assert open(tmp_path / "w.tsv", "rb").read() == open(os.path.join(C1, "clust_workflow.tsv"), "rb").read()
    util.tsv_invariants(os.path.join(C1, "clust_workflow.tsv"), names)


@pytest.mark.gpu
def test_c1_example_data_hip_path(tmp_path):
    import unicore_amd as U
    db = os.path.join(C1, "db")
    U.cluster(db, str(tmp_path / "c_cluster"), str(tmp_path / "tmp"), "-c 0.8 --single-step-clustering")
    U.createtsv(db, str(tmp_path / "c_cluster"), str(tmp_path / "c.tsv"))
    assert open(tmp_path / "c.tsv", "rb").read() == open(os.path.join(C1, "clust.tsv"), "rb").read()
    U.cluster(db, str(tmp_path / "w_cluster"), str(tmp_path / "tmp"), "-c 0.8")      # the default workflow (pre-step + cascade)
    U.createtsv(db, str(tmp_path / "w_cluster"), str(tmp_path / "w.tsv"))
    assert open(tmp_path / "w.tsv", "rb").read() == open(os.path.join(C1, "clust_workflow.tsv"), "rb").read()


@pytest.mark.gpu
def test_real_foldseek_conformance(tmp_path):
    """if a real `foldseek` is on PATH, run `cluster --single-step-clustering` + `createtsv` on the C1
    fixture and diff clust.tsv against this engine.  No Foldseek exists in this image or on the GPU box, so this always
    skips - which is exactly why the oracle says PARITY UNPINNED."""
    import shutil
    import subprocess
    fs = shutil.which("foldseek")
    ours = os.path.realpath(os.path.join(util.ROOT, "bin", "foldseek"))
    if fs is None or os.path.realpath(fs) == ours:
        pytest.skip("no real foldseek on PATH: parity against the third-party binary stays unpinned")
    import unicore_amd as U
    db = os.path.join(C1, "db")
    subprocess.check_call([fs, "cluster", "--threads", "4", "-v", "1", db, str(tmp_path / "f_cluster"), str(tmp_path / "ftmp"), "-c", "0.8",
                           "--single-step-clustering"])
    subprocess.check_call([fs, "createtsv", "--threads", "4", "-v", "1", db, db, str(tmp_path / "f_cluster"), str(tmp_path / "f.tsv")])
    U.cluster(db, str(tmp_path / "c_cluster"), str(tmp_path / "tmp"), "-c 0.8 --single-step-clustering")
    U.createtsv(db, str(tmp_path / "c_cluster"), str(tmp_path / "c.tsv"))
    assert open(tmp_path / "c.tsv", "rb").read() == open(tmp_path / "f.tsv", "rb").read(), \
        "clust.tsv differs from real Foldseek: the spec's EXT-UNVERIFIED constants (matrix, thresholds, E-value model) need the real data files"
