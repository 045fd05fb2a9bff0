#!/usr/bin/env python3
"""tools/oracle_at_size.py --config c3 [--threads N] [--chunk Q] [--work DIR] [--out tests/golden/c3_sha.json]

The CPU oracle END TO END at a BASELINE configuration's full size (north_star's "byte-identical clust.tsv
on 500 proteomes at 1 GPU" needs a whole-TSV answer from the CPU side, not a query sample).  Runs in the BUILD container (no GPU):
the plain all-vs-all step of spec UC-1.1 — E1 index, E2-E4 per query (the scalar oracle's prefilter), E5/E6 through the AVX2
inter-sequence leg `oracle/uc_simd.c` (record for record equal to the scalar oracle: tests/test_oracle_kat.py), E7 `uco_setcover`,
E9 `uco_write_tsv` — in query chunks with a checkpoint per chunk (a killed run resumes), and writes

    {sha256 of clust.tsv, bytes, clusters, every stage counter of uco_cluster, per-chunk edge checksum}

as a small golden fixture.  `tests/test_configs_gpu.py::test_config_at_size[c3]` asserts the HIP path's TSV hash and counters
against it on the GPU box.  `--selfcheck` compares this chunked driver with `uco_cluster` itself on a small database first."""
import argparse, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import numpy as np
import util
from oracle import oracle_py as O

CONFIGS = {  # == bench.py CONFIGS / tests/test_configs_gpu.py / tests/test_workflow_gpu.py (name: proteomes, families, scale, seed, options, target sensitivity)
    "c2": (50, 6000, 1.0, 0x5EED0002, "-c 0.8", 4.0),
    "c3": (500, 6000, 1.0, 0x5EED0003, "-c 0.8", 4.0),
    "c4-lite": (50, 6000, 1.0, 0x5EED0004, "-c 0.8 --min-seq-id 0.3 -s 7.5", 7.5),      # configs[3]'s options, PLAIN step, 50 proteomes
    "c4-200": (200, 6000, 1.0, 0x5EED0004, "-c 0.8 --min-seq-id 0.3 -s 7.5", 7.5),
    "c4-500": (500, 6000, 1.0, 0x5EED0004, "-c 0.8 --min-seq-id 0.3 -s 7.5", 7.5),
    "c4-1000": (1000, 6000, 1.0, 0x5EED0004, "-c 0.8 --min-seq-id 0.3 -s 7.5", 7.5),   # r06: the largest size whose end-to-end CPU run fits a round (nominal 2000: ~100 core-hours)
    "c3-gate": (500, 6000, 1.0, 0x5EED0003, "-c 0.8 --length-gate 1", 4.0),      # optional rule UC-1/L on (default off)
    "c2-gate": (50, 6000, 1.0, 0x5EED0002, "-c 0.8 --length-gate 1", 4.0),
}


def covered(p, lq, lt):
    """rule UC-1/L as the C side evaluates it (float32 division): which pairs the length gate lets through; all of them when the rule is off"""
    lq, lt = np.broadcast_arrays(np.asarray(lq, np.int64), np.asarray(lt, np.int64))
    if not p.len_gate or not p.cov > 0:
        return np.ones(lq.shape, bool)
    a, b, c = lq.astype(np.float32), lt.astype(np.float32), np.float32(p.cov)
    with np.errstate(divide="ignore", invalid="ignore"):
        r1, r2 = a / b >= c, b / a >= c
    return ((r1 & r2) if p.cov_mode == 0 else r1 if p.cov_mode == 1 else r2) & (lq > 0) & (lt > 0)


def run_chunked(odb, p, threads, chunk, work, log=None):
    """-> (assign, counts) of the plain step, computed chunk by chunk (checkpointed under `work` if given)"""
    n = odb.n
    off = odb.offsets().astype(np.int64)
    lens = off[1:] - off[:-1]
    dbres = int(off[n])
    ix = None
    tot = dict(n_sim_kmers=0, n_kmer_hits=0, n_candidates=0, n_prefilter_hits=0, n_alignments=0, n_edges=0,
               cells_fwd=0, cells_rev=0, cells_start=0)
    edge_parts, t_pre, t_aln = [], 0.0, 0.0
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        ck = os.path.join(work, "chunk_%09d_%09d.npz" % (c0, c1)) if work else None
        if ck and os.path.exists(ck):
            z = np.load(ck)
            e, cn = z["edges"], json.loads(str(z["counts"]))
        else:
            if ix is None:
                t = time.perf_counter()
                ix = O.build_index(odb, p)
                if log: log("index built in %.1f s" % (time.perf_counter() - t))
            q = np.arange(c0, c1, dtype=np.uint32)
            npairs, s0, s1, cnt, hits, alns, pc = O.simd_run_counts(odb, ix, p, q, threads=threads)
            t_pre += s0; t_aln += s1
            M = hits.shape[1]
            valid = np.arange(M)[None, :] < cnt[:, None]
            lq = lens[c0:c1][:, None]
            lt = lens[hits["t"]] * valid
            ms = np.array([O.lib().uco_min_score(p, int(l), dbres) for l in lens[c0:c1]], np.int64)[:, None]
            acc = valid & (alns["accepted"] == 1)
            qq = np.broadcast_to(q[:, None], acc.shape)
            e = np.stack([qq[acc], hits["t"][acc]], axis=1).astype(np.uint32)      # (q asc, list order) == uco_cluster's edge order
            cn = dict(pc)
            cov = valid & covered(p, lq, lens[hits["t"]])          # gated pairs (rule UC-1/L) are no alignments and run no cells
            cn["n_alignments"] = int(cov.sum()); cn["n_edges"] = int(len(e))
            cn["cells_fwd"] = int((lq * lt * cov).sum())
            cn["cells_rev"] = int((lq * lt * (cov & (alns["score"] >= ms))).sum()) if p.rev_correction else 0
            cn["cells_start"] = int((((alns["qend"].astype(np.int64) + 1) * (alns["tend"].astype(np.int64) + 1)) * (valid & (alns["pass_evalue"] == 1))).sum())
            assert npairs == cn["n_alignments"]
            if ck:
                np.savez(ck + ".tmp.npz", edges=e, counts=json.dumps(cn))
                os.replace(ck + ".tmp.npz", ck)
            if log: log("queries [%d, %d): %d alignments, %d edges, prefilter %.0f s, gapped %.0f s" % (c0, c1, cn["n_alignments"], len(e), s0, s1))
        edge_parts.append(e)
        for k in tot:
            tot[k] += int(cn.get(k, 0))
    if ix is not None:
        O.free_index(ix)
    edges = np.concatenate(edge_parts) if edge_parts else np.zeros((0, 2), np.uint32)
    assign = O.setcover(n, edges)
    tot["n_clusters"] = int((assign == np.arange(n)).sum())
    # the accepted pairs as a SET: sha256 over the sorted (query << 32 | target) keys (the engine appends its edges in plan order)
    key = np.sort(edges[:, 0].astype(np.uint64) << np.uint64(32) | edges[:, 1].astype(np.uint64))
    tot["edge_set_sha256"] = hashlib.sha256(key.tobytes()).hexdigest()
    return assign, tot, (t_pre, t_aln)


def run_workflow(odb, p, target_s, steps, m, threads, chunk, work, log=None):
    """the DEFAULT workflow (uco_cluster_workflow: linear-time pre-step + `steps` cascade rounds on the representatives, sensitivity 1 -> target_s)
    with the gapped stages through the SIMD leg and the cascade rounds chunked / checkpointed -> (assign, counts summed over the rounds, round sizes)"""
    n = odb.n
    thr = O.cascade_thresholds(p, target_s, steps)
    cur = np.arange(n, dtype=np.int64)
    assign = np.arange(n, dtype=np.int64)
    total = dict(n_sim_kmers=0, n_kmer_hits=0, n_candidates=0, n_prefilter_hits=0, n_alignments=0, n_edges=0, cells_fwd=0, cells_rev=0, cells_start=0)
    sizes = []
    for r in range(steps + 1):
        sizes.append(int(len(cur)))
        sub = odb if len(cur) == n else odb.subset(cur)
        if r == 0:                                   # E8a: (centre, member) pairs, every pair through E5/E6, set cover
            pairs = O.linclust_pairs(sub, p, m)
            al = O.simd_align_pairs(sub, p, pairs, threads=threads)
            off = sub.offsets().astype(np.int64); lens = off[1:] - off[:-1]
            lq, lt = lens[pairs[:, 0]], lens[pairs[:, 1]]
            ms = np.array([O.lib().uco_min_score(p, int(l), int(off[-1])) for l in np.unique(lq)], np.int64)
            msq = ms[np.searchsorted(np.unique(lq), lq)]
            cov = covered(p, lq, lt)
            c = dict(n_prefilter_hits=len(pairs), n_alignments=int(cov.sum()), n_edges=int((al["accepted"] == 1).sum()),
                     cells_fwd=int((lq * lt * cov).sum()), cells_rev=int((lq * lt * (cov & (al["score"] >= msq))).sum()) if p.rev_correction else 0,
                     cells_start=int(((al["qend"].astype(np.int64) + 1) * (al["tend"].astype(np.int64) + 1) * (al["pass_evalue"] == 1)).sum()))
            sa_ = O.setcover(sub.n, pairs[al["accepted"] == 1]).astype(np.int64)
            if log: log("pre-step: %d sequences, %d pairs, %d accepted" % (sub.n, len(pairs), c["n_edges"]))
        else:
            keep = p.kmer_thr
            p.kmer_thr = thr[r - 1]
            w = os.path.join(work, "round%d" % r) if work else None
            if w: os.makedirs(w, exist_ok=True)
            sa_, c, _ = run_chunked(sub, p, threads, chunk, w, log)
            p.kmer_thr = keep
            sa_ = sa_.astype(np.int64)
            if log: log("round %d: %d sequences, k-score %d, %d alignments, %d accepted" % (r, sub.n, thr[r - 1], c["n_alignments"], c["n_edges"]))
        for k in total:
            total[k] += int(c.get(k, 0))
        posmap = np.zeros(n, np.int64); posmap[cur] = np.arange(len(cur))
        assign = cur[sa_[posmap[assign]]]              # mergeclusters: the representative of a sequence is the representative of its representative
        cur = cur[sa_ == np.arange(len(cur))]
        del sub
    total["n_clusters"] = int(len(cur))
    return assign.astype(np.uint32), total, sizes


def selfcheck():
    """the chunked driver == uco_cluster (assignment and all counters) on a family database"""
    s3, sa = util.family_db(11, n_fam=40, members=6, extra=(700, 900))
    odb = O.OracleDb(s3=s3, sa=sa)
    for opts in ("-c 0.8", "-c 0.8 --length-gate 1", "-c 0.7 --cov-mode 2 --length-gate 1"):       # the last two: optional rule UC-1/L
        p = util.oracle_params(O, opts)
        ref = O.cluster(odb, p, threads=4, dumps=False)
        assign, tot, _ = run_chunked(odb, p, 4, 37, None)
        assert np.array_equal(assign, ref["assign"])
        for k, v in ref["counts"].items():
            assert int(v) == tot[k], (opts, k, int(v), tot[k])
    # ... and the workflow driver == uco_cluster_workflow on a synthetic proteome set
    import tempfile
    d = tempfile.mkdtemp(prefix="uc_oas_")
    db = util.gen_synth_db(os.path.join(d, "db"), 6, 0x5EED0004, 40, 0.6)
    odb = O.OracleDb(db)
    for opts, s_ in (("-c 0.8", 4.0), ("-c 0.8 --min-seq-id 0.3 -s 7.5", 7.5), ("-c 0.8 --length-gate 1", 4.0)):
        p = util.oracle_params(O, opts)
        ref = O.cluster_workflow(odb, p, O.cascade_thresholds(p, s_, 3), linclust_m=20, threads=4)
        assign, tot, sizes = run_workflow(odb, p, s_, 3, 20, 4, 50, None)
        assert np.array_equal(assign, ref["assign"]) and sizes == [int(x) for x in ref["round_sizes"]]
        for k in ("n_alignments", "n_edges", "n_clusters", "n_prefilter_hits", "cells_fwd", "cells_rev", "cells_start", "n_kmer_hits", "n_candidates", "n_sim_kmers"):
            assert int(ref["counts"][k]) == tot[k], (opts, k, int(ref["counts"][k]), tot[k])
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--threads", type=int, default=max(1, len(os.sched_getaffinity(0)) - 1))
    ap.add_argument("--chunk", type=int, default=20000)
    ap.add_argument("--work", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--selfcheck", action="store_true")
    ap.add_argument("--workflow", action="store_true", help="the DEFAULT workflow (pre-step + 3-step cascade) instead of the plain step -> <config>_workflow_sha.json")
    a = ap.parse_args()
    if a.selfcheck:
        print("selfcheck", selfcheck()); return
    prot, fam, scale, seed, opts, target_s = CONFIGS[a.config]
    work = a.work or "/tmp/uc_oracle_%s" % a.config
    os.makedirs(work, exist_ok=True)
    t0 = time.perf_counter()
    def log(m): print("[%7.0f s] %s" % (time.perf_counter() - t0, m), file=sys.stderr, flush=True)
    db = os.path.join(work, "db")
    if not os.path.exists(db + ".map"):
        util.gen_synth_db(db, prot, seed, fam, scale)
    odb = O.OracleDb(db)
    p = util.oracle_params(O, opts)
    log("%d sequences, %d residues, options %r, %d threads" % (odb.n, int(odb.offsets()[-1]), opts, a.threads))
    sizes = None
    if a.workflow:
        assign, tot, sizes = run_workflow(odb, p, target_s, 3, 20, a.threads, a.chunk, os.path.join(work, "workflow"), log)
        tp = ta = 0.0
    else:
        assign, tot, (tp, ta) = run_chunked(odb, p, a.threads, a.chunk, work, log)
    tsv = os.path.join(work, "clust_workflow.tsv" if a.workflow else "clust.tsv")
    O.write_tsv(tsv, odb, assign)
    data = open(tsv, "rb").read()
    res = {"config": a.config, "proteomes": prot, "seed": hex(seed), "options": opts + ("" if a.workflow else " --single-step-clustering"),
           "workflow": "default (pre-step + 3-step cascade)" if a.workflow else "plain step", "round_sizes": sizes,
           "sequences": int(odb.n), "residues": int(odb.offsets()[-1]),
           "tsv_sha256": hashlib.sha256(data).hexdigest(), "tsv_bytes": len(data), "counts": tot,
           "made_by": "tools/oracle_at_size.py (build container, %d threads; prefilter %.0f s + gapped %.0f s of this process, checkpointed chunks not included)" % (a.threads, tp, ta)}
    out = a.out or os.path.join(ROOT, "tests", "golden", "%s%s_sha.json" % (a.config, "_workflow" if a.workflow else ""))
    json.dump(res, open(out, "w"), indent=1)
    log("wrote " + out)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
