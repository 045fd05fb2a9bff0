#!/usr/bin/env python3
"""tools/property_campaign.py <first_seed> <n_seeds> — the property test of tests/test_gpu_parity.py (random small databases x
random option strings against the oracle: hit lists, every alignment field, set cover) for seeds beyond the 16 of the suite;
every third seed also forces the prefilter's super-batch cut (UC_DRUN_MAX), every fourth tiny index chunks (the upper-triangle walk of
the chunk grid), every fifth a narrow traceback band (UC_TB_BAND 1 / 2 / 4 / 8: the whole-box fallback) and every seventh a 1 MB matrix
budget (one task per batch, 1000-pair plan segments).  Run on the GPU box after
kernel changes; the result line goes into DESIGN.md 2."""
import os, sys, time, traceback
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import torch  # noqa: F401
from oracle import oracle_py as O
import test_gpu_parity as T
first, n = int(sys.argv[1]), int(sys.argv[2])
fn = T.test_random_option_sets_against_the_oracle
fn = getattr(fn, "__wrapped__", fn)
bad, t0 = [], time.time()
for seed in range(first, first + n):
    env = {}
    if seed % 3 == 1: env = {"UC_DRUN_MAX": str(500 + 37 * (seed % 50))}
    if seed % 4 == 3: env = dict(env, UC_PREFILTER_CHUNK_RES=str(1500 + 113 * (seed % 40)))      # r04: several index chunks -> the symmetric (upper-triangle) walk + the single merge
    if seed % 5 == 2: env = dict(env, UC_TB_BAND=str((1, 2, 4, 8)[seed % 4]))                    # r05: narrow traceback bands -> the whole-box fallback
    if seed % 7 == 4: env = dict(env, UC_TB_BUDGET_MB="1")                                       # r05: many matrix batches, plan segments
    os.environ.update(env)
    try:
        fn(O, seed)
    except Exception as e:   # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", "".join(traceback.format_exception_only(type(e), e)).strip()[:400], flush=True)
    finally:
        for k in env: os.environ.pop(k, None)
print("property campaign: seeds %d..%d, %d failures %s, %.0f s" % (first, first + n - 1, len(bad), bad, time.time() - t0))

# ---- workflows: random synthetic proteome sets through uc_cluster (pre-step on/off, 1-3 cascade steps, coverage, k-mers per
#      sequence) against uco_cluster_workflow: clust.tsv bytes, cluster and alignment counts
import numpy as np, shutil, tempfile
import unicore_amd as U, util
nw = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, n // 6)
badw, t0 = [], time.time()
for seed in range(first, first + nw):
    rng = np.random.default_rng(7000 + seed)
    d = tempfile.mkdtemp(prefix="uc_wf_")
    try:
        db = util.gen_synth_db(os.path.join(d, "db"), int(rng.integers(3, 9)), 0x5EED1000 + seed, int(rng.integers(20, 80)), float(rng.choice([0.4, 0.6, 1.0])))
        steps, lin, m = int(rng.integers(1, 4)), int(rng.integers(0, 2)), int(rng.choice([3, 5, 20, 40]))
        base = "-c %.1f --cov-mode %d" % (rng.choice([0.5, 0.8, 0.9]), rng.integers(0, 3))
        if rng.random() < 0.3: base += " --min-seq-id %.1f" % rng.choice([0.3, 0.5])
        if rng.random() < 0.3: base += " --length-gate 1"      # optional rule UC-1/L through every round of the workflow
        opts = base + " --linclust %d --cluster-steps %d" % (lin, steps) + (" --kmer-per-seq %d" % m if lin else "")
        if steps == 1 and not lin: opts = base + " --single-step-clustering"
        st = U.cluster(db, os.path.join(d, "c_cluster"), os.path.join(d, "tmp"), opts, threads=4)
        U.createtsv(db, os.path.join(d, "c_cluster"), os.path.join(d, "c.tsv"))
        odb = O.OracleDb(db)
        p = util.oracle_params(O, base)
        ref = O.cluster_workflow(odb, p, O.cascade_thresholds(p, 4.0, steps), linclust_m=m if lin else 0, threads=8)
        O.write_tsv(os.path.join(d, "r.tsv"), odb, ref["assign"])
        same = open(os.path.join(d, "c.tsv"), "rb").read() == open(os.path.join(d, "r.tsv"), "rb").read()
        if not (same and st["n_clusters"] == ref["counts"]["n_clusters"] and st["n_gapped_alignments"] == ref["counts"]["n_alignments"]):
            badw.append(seed); print("workflow seed", seed, "FAILED:", opts, st["n_clusters"], ref["counts"]["n_clusters"], st["n_gapped_alignments"], ref["counts"]["n_alignments"], flush=True)
    except Exception as e:   # noqa: BLE001
        badw.append(seed); print("workflow seed", seed, "FAILED:", "".join(traceback.format_exception_only(type(e), e)).strip()[:400], flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
print("workflow campaign: seeds %d..%d, %d failures %s, %.0f s" % (first, first + nw - 1, len(badw), badw, time.time() - t0))
