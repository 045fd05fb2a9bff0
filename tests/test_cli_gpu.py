"""The reference boundary under test, end to end, on the GPU: the argv of /root/reference/src/modules/cluster.rs:45-73
against bin/foldseek (the drop-in an unmodified Unicore is pointed at through path.cfg), `bin/unicore cluster [-k]`
(the C++ mirror of modules::cluster::run), and the multi-GPU layout behind the same C entry point run
as several virtual ranks on the one GPU of the test box."""
import os
import subprocess

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

ROOT = util.ROOT
GOLD = os.path.join(ROOT, "tests", "golden")
SHIM = os.path.join(ROOT, "bin", "foldseek")
EXE = os.path.join(ROOT, "bin", "unicore")

# committed golden TSVs per database: what a bare "-c 0.8" (Foldseek's default workflow: pre-step + 3-step cascade) and
# what "--single-step-clustering" must produce
CASES = {
    "db": (os.path.join(GOLD, "db"), os.path.join(GOLD, "clust_linclust_cascade3.tsv"), os.path.join(GOLD, "clust_default.tsv")),
    "c1": (os.path.join(GOLD, "c1", "db"), os.path.join(GOLD, "c1", "clust_workflow.tsv"), os.path.join(GOLD, "c1", "clust.tsv")),
}


def run(argv, **kw):
    r = subprocess.run(argv, capture_output=True, text=True, **kw)
    assert r.returncode == 0, (argv, r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return r


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("extra", [[], ["--single-step-clustering"]])
def test_foldseek_shim_with_the_reference_argv(case, extra, tmp_path):
    """cluster.rs:45-49 -> :59-62 -> :67-73, token for token (threads 4, verbosity 2, options AFTER the positionals)"""
    db, gold_default, gold_single = CASES[case]
    out = str(tmp_path / "res" / "clust")
    os.makedirs(os.path.dirname(out))
    tmp = str(tmp_path / "tmp")            # the callee creates <tmp>
    run([SHIM, "cluster", "--threads", "4", "-v", "2", db, out + "_cluster", tmp, "-c", "0.8"] + extra)
    for sfx in ("", ".index", ".dbtype"):
        assert os.path.exists(out + "_cluster" + sfx)
    assert os.path.isdir(tmp)
    run([SHIM, "createtsv", "--threads", "4", "-v", "2", db, db, out + "_cluster", out + ".tsv"])
    assert open(out + ".tsv", "rb").read() == open(gold_single if extra else gold_default, "rb").read()
    run([SHIM, "rmdb", out + "_cluster", "-v", "2"])
    assert sorted(os.listdir(os.path.dirname(out))) == ["clust.tsv"]       # nothing but <out>.tsv is left beside it


def test_foldseek_shim_flags_before_the_positionals(tmp_path):
    db, _, gold_single = CASES["db"]
    out = str(tmp_path / "clust")
    run([SHIM, "cluster", "-c", "0.8", "--single-step-clustering", db, "--threads", "2", out + "_cluster", "-v", "1", str(tmp_path / "tmp")])
    run([SHIM, "createtsv", db, db, out + "_cluster", out + ".tsv", "--threads", "2"])
    assert open(out + ".tsv", "rb").read() == open(gold_single, "rb").read()


@pytest.mark.parametrize("keep", [False, True])
def test_unicore_cluster_success_path(keep, tmp_path):
    """modules::cluster::run (cluster.rs:9-84): OUT.tsv written, cluster.chk "0" -> "1" (cluster.rs:32,81), OUT_cluster*
    removed unless -k (cluster.rs:67-76), missing parent directory created (cluster.rs:27-29)"""
    db, gold_default, _ = CASES["c1"]
    out = str(tmp_path / "newdir" / "sub" / "clust")
    argv = [EXE, "cluster", db, out, str(tmp_path / "tmp"), "--threads", "4"] + (["-k"] if keep else [])
    r = run(argv)
    assert "Running cluster" in r.stdout and " Done" in r.stdout               # message.rs / cluster.rs:52,56 at verbosity 3
    parent = os.path.dirname(out)
    assert open(os.path.join(parent, "cluster.chk")).read() == "1"
    assert open(out + ".tsv", "rb").read() == open(gold_default, "rb").read()
    left = sorted(os.listdir(parent))
    if keep:
        assert left == ["clust.tsv", "clust_cluster", "clust_cluster.dbtype", "clust_cluster.index", "cluster.chk"]
        assert int.from_bytes(open(out + "_cluster.dbtype", "rb").read(), "little") == 6
    else:
        assert left == ["clust.tsv", "cluster.chk"]
    names = [l.split("\t")[1] for l in open(db + ".lookup")]
    util.tsv_invariants(out + ".tsv", names)


def test_unicore_cluster_options_string_and_quiet(tmp_path):
    db, _, gold_single = CASES["c1"]
    out = str(tmp_path / "clust")
    r = run([EXE, "cluster", db, out, str(tmp_path / "tmp"), "-c", "-c 0.8 --single-step-clustering", "-v", "0"], cwd=str(tmp_path))
    assert r.stdout == "" and r.stderr == ""                                        # verbosity 0: quiet
    assert open(out + ".tsv", "rb").read() == open(gold_single, "rb").read()
    assert open(str(tmp_path / "cluster.chk")).read() == "1"


def test_out_of_memory_relief_and_retry(tmp_path):
    """uc_engine.h malloc_with_relief / Engine::relieve_pressure: a device allocation that fails with out-of-memory inside a stage makes the
    engine give back what that stage is not using (the other stage's work buffers, parked buffers) and is tried once more.  The failure is
    injected (UC_TEST_OOM_AT=k: the k-th allocation under a registered handler reports out-of-memory once) at allocations spread over the
    default workflow - uploads, prefilter, gapped stage, later rounds where both stages hold scratch - and the result may not change a byte."""
    db, gold_default, _ = CASES["c1"]
    fired, stages = 0, set()
    for k in (1, 2, 3, 5, 8, 13, 21, 34, 55, 89):
        out = str(tmp_path / ("k%d" % k) / "clust")
        r = run([EXE, "cluster", db, out, str(tmp_path / "tmp"), "--threads", "4"], env=dict(os.environ, UC_TEST_OOM_AT=str(k)))
        assert open(out + ".tsv", "rb").read() == open(gold_default, "rb").read(), k
        msg = [l for l in (r.stdout + r.stderr).splitlines() if "device memory ran out in the" in l]
        assert len(msg) <= 1, msg
        if msg:
            fired += 1
            stages.add(msg[0].split("ran out in the ")[1].split(";")[0])
    assert fired >= 6 and len(stages) >= 2, (fired, stages)      # the handler ran in most runs and for more than one stage


def test_unicore_search_success_path(tmp_path):
    """modules::search::run (search.rs:8-84) through bin/unicore: OUT.m8 equals the committed golden, search.chk == "1" """
    db = CASES["db"][0]
    out = str(tmp_path / "hits")
    run([EXE, "search", db, db, out, str(tmp_path / "tmp"), "-v", "1"])
    assert open(out + ".m8", "rb").read() == open(os.path.join(GOLD, "search_self.m8"), "rb").read()
    assert open(str(tmp_path / "search.chk")).read() == "1"
    assert not os.path.exists(out + "_aln")


# ---------------------------------------------------------------------------------------------- multi-GPU behind the C ABI
@pytest.fixture(scope="module")
def synth_db(tmp_path_factory):
    d = tmp_path_factory.mktemp("mg")
    return util.gen_synth_db(str(d / "db"), 8, 0x5EED0007, 60, 0.7)


def _tsv(db, tmp_path, tag, opts, num_gpus, env=None):
    import unicore_amd as U
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        st = U.cluster(db, str(tmp_path / (tag + "_cluster")), str(tmp_path / "tmp"), opts, threads=4, num_gpus=num_gpus)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    U.createtsv(db, str(tmp_path / (tag + "_cluster")), str(tmp_path / (tag + ".tsv")))
    return open(tmp_path / (tag + ".tsv"), "rb").read(), st


@pytest.mark.parametrize("workflow", ["-c 0.8 --single-step-clustering", "-c 0.8", "-c 0.8 --min-seq-id 0.3 --single-step-clustering --max-seqs 7"])
def test_uc_cluster_virtual_gpus_give_identical_tsv(synth_db, workflow, tmp_path):
    """SURVEY.md 8e determinism: the result must not depend on the number of GPUs or on the Q x T grid.  uc_cluster with
    2 / 3 / 4 ranks (one engine + one host thread each; several ranks per physical device, so the exchange uses device
    copies instead of RCCL) against the single-GPU run: clust.tsv bytes, alignments and cluster count identical."""
    ref, st1 = _tsv(synth_db, tmp_path, "g1", workflow, 1)
    assert st1["n_gpus"] == 1
    for tag, opts, n in (("g2", workflow, 2),                                     # T = 2: the north-star target partition
                         ("g4", workflow + " --target-shards 2", 4),              # Q2 x T2
                         ("g3", workflow + " --target-shards 1", 3)):             # Q3 x T1: query groups
        got, st = _tsv(synth_db, tmp_path, tag, opts, n, env={"UC_VIRTUAL_GPUS": "1"})
        assert got == ref, (tag, opts)
        assert st["n_gpus"] == n and st["n_clusters"] == st1["n_clusters"]
        assert st["n_gapped_alignments"] == st1["n_gapped_alignments"], tag      # every merged pair aligned exactly once over all ranks
        assert st["exchange_bytes"] > 0


def test_uc_cluster_virtual_gpus_shard_by_shard_exchange(synth_db, tmp_path):
    """a home rank that would receive more than UC_EXCHANGE_LIMIT records at once works its query range off in several rounds
    (sub-ranges of the home ranges, merged round by round): same result"""
    ref, _ = _tsv(synth_db, tmp_path, "g1", "-c 0.8 --single-step-clustering", 1)
    got, _ = _tsv(synth_db, tmp_path, "g3", "-c 0.8 --single-step-clustering", 3, env={"UC_VIRTUAL_GPUS": "1", "UC_EXCHANGE_LIMIT": "10"})
    assert got == ref


@pytest.mark.parametrize("opts", ["-c 0.8 --single-step-clustering", "-c 0.8 --single-step-clustering -s 7.5"])
def test_prefilter_execution_variants_give_identical_tsv(synth_db, opts, tmp_path):
    """E2 runs the similar-k-mer enumeration once per DISTINCT query k-mer and plans its batches from exact per-query totals.
    The cut into query super-batches (forced here by a tiny run-list budget) may not change a byte of the result or any of the prefilter counts."""
    ref, st1 = _tsv(synth_db, tmp_path, "v0", opts, 1)
    keys = ("n_sim_kmers", "n_kmer_hits", "n_filtered_hits", "n_candidates", "n_prefilter_hits", "n_gapped_alignments", "n_clusters")
    # ... nor may the cut of the targets into index chunks — by default walked as the UPPER TRIANGLE of the chunk x chunk grid, the pairs the other
    # way round mirrored from the symmetric hit relation (fewer keys reach the sort, every other count is the algorithm's), with
    # UC_PREFILTER_SYMMETRIC=0 as the full grid
    small = {"UC_PREFILTER_CHUNK_RES": "20000"}
    tiny = {"UC_PREFILTER_CHUNK_RES": "6000"}
    for tag, env in (("v1", {"UC_DRUN_MAX": "20000"}), ("v2", {"UC_DRUN_MAX": "3000"}),
                     ("v4", small), ("v5", tiny), ("v6", dict(small, UC_PREFILTER_SYMMETRIC="0")), ("v7", dict(small, UC_DRUN_MAX="20000")),
                     ("v8", dict(tiny, UC_PREFILTER_WIDE="1"))):
        got, st = _tsv(synth_db, tmp_path, tag, opts, 1, env=env)
        assert got == ref, (tag, opts)
        kk = [k for k in keys if not (tag in ("v4", "v5", "v7", "v8") and k == "n_filtered_hits")]      # the triangle walk expands (and sorts) fewer keys
        assert {k: st[k] for k in kk} == {k: st1[k] for k in kk}, tag
        if tag in ("v4", "v5"):
            assert st["n_filtered_hits"] < st1["n_filtered_hits"], tag


def test_more_gpus_than_visible_is_an_error_not_a_silent_fallback(synth_db, tmp_path):
    import unicore_amd as U
    with pytest.raises(U.UcError) as ei:
        U.cluster(synth_db, str(tmp_path / "x_cluster"), str(tmp_path / "tmp"), "-c 0.8", num_gpus=64)
    assert ei.value.code == U.UC_ERR_DEVICE and "visible" in str(ei.value)
    with pytest.raises(U.UcError):
        _tsv(synth_db, tmp_path, "bad", "-c 0.8 --target-shards 3", 4, env={"UC_VIRTUAL_GPUS": "1"})      # 3 does not divide 4


def test_shim_uses_every_rank_it_is_given(synth_db, tmp_path):
    """`foldseek cluster ... --gpus 2` (virtual ranks here) == the single-GPU run, through the executable"""
    out = str(tmp_path / "clust")
    env = dict(os.environ, UC_VIRTUAL_GPUS="1")
    r = run([SHIM, "cluster", "--threads", "4", "-v", "3", synth_db, out + "_cluster", str(tmp_path / "tmp"), "-c", "0.8", "--gpus", "2"], env=env)
    assert "2 GPU(s)" in r.stdout and "1 query group(s) x 2 target shard(s)" in r.stdout
    run([SHIM, "createtsv", synth_db, synth_db, out + "_cluster", out + ".tsv"])
    ref, _ = _tsv(synth_db, tmp_path, "g1", "-c 0.8", 1)
    assert open(out + ".tsv", "rb").read() == ref


def test_cluster_step_and_one_rank_rccl_communicator(synth_db):
    """uc_engine_cluster_step: (a) comm = NULL equals the staged calls; (b) with a 1-rank RCCL communicator the pass goes
    through the RCCL entry points of an N-rank run (ncclCommInitRank, ncclAllGather of the count matrices, the grouped
    point-to-point exchanges — empty with one rank: its own slices are device copies — and the edge gather) and still
    reproduces it; RCCL itself reports the communicator's rank count."""
    import unicore_amd as U
    e = U.Engine("-c 0.8", threads=4)
    e.load_db(synth_db)
    e.prefilter()
    e.align()
    ref = e.setcover(e.edges())
    n_ref = e.hits_size()
    a0, k0 = e.cluster_step()
    assert np.array_equal(a0, ref) and k0 == n_ref
    comm = U.Comm(U.Comm.unique_id(), 0, 1)
    a1, k1 = e.cluster_step(comm, target_shards=0)
    assert np.array_equal(a1, ref) and k1 == n_ref
    st = e.stats()
    assert st["exchange_seconds"] > 0 and st["exchange_bytes"] == 0          # bytes received from PEERS: none with one rank
    assert comm.info()[:2] == (1, 0)
    # a failure BETWEEN ncclGroupStart and ncclGroupEnd must close the group on its way out (scope guard) — the same communicator
    # and thread then still work.  UC_FAIL_RANK=<rank>:2 throws inside the grouped exchange.
    os.environ["UC_FAIL_RANK"] = "0:2"
    try:
        with pytest.raises(U.UcError) as ei:
            e.cluster_step(comm, target_shards=0)
        assert "inside the grouped exchange" in str(ei.value)
    finally:
        del os.environ["UC_FAIL_RANK"]
    a2, k2 = e.cluster_step(comm, target_shards=0)
    assert np.array_equal(a2, ref) and k2 == n_ref
    comm.close()
    e.close()
    U.lib().uc_release_scratch()
