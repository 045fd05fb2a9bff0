#!/usr/bin/env python3
"""tools/widen_bench.py — timings of the widened rows on the bench database (GPU box):
uc_cluster single step vs 3-step cascade, and uc_search (a 5-proteome query DB against the 50-proteome DB) +
uc_convertalis.  From-disk wall clock (read + encode + upload included), stats from the C ABI."""
import os, sys, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import torch  # noqa: F401
import bench, unicore_amd as U
wd = "/tmp/uc_bench/p50"
bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002)
db = os.path.join(wd, "db")
wq = "/tmp/uc_bench/p5q"
bench.gen_db(wq, 5, 6000, 1.0, 0x5EED0002)    # same families as the target DB (its first 5 proteomes)
qdb = os.path.join(wq, "db")
out = {}
for name, opts in (("cluster_single", "-c 0.8 --single-step-clustering"), ("cluster_cascade3", "-c 0.8 --linclust 0 --cluster-steps 3"), ("cluster_default_workflow", "-c 0.8"),
                   ("cluster_seqid", "-c 0.8 --single-step-clustering --min-seq-id 0.3")):
    U.cluster(db, "/tmp/uc_bench/w_cluster", "/tmp/uc_bench/tmp", opts)          # warm-up (allocations, page cache)
    t = time.perf_counter()
    st = U.cluster(db, "/tmp/uc_bench/w_cluster", "/tmp/uc_bench/tmp", opts)
    dt = time.perf_counter() - t
    out[name] = {"options": opts, "wall_s": round(dt, 3), "alignments": st["n_gapped_alignments"], "clusters": st["n_clusters"],
                 "sw_kernel_ms": round(st["sw_kernel_ms"], 1), "prefilter_kernel_ms": round(st["prefilter_kernel_ms"], 1),
                 "alignments_per_s": round(st["n_gapped_alignments"] / dt)}
U.search(qdb, db, "/tmp/uc_bench/w_aln", "/tmp/uc_bench/tmp", "-c 0.8")
t = time.perf_counter()
st = U.search(qdb, db, "/tmp/uc_bench/w_aln", "/tmp/uc_bench/tmp", "-c 0.8")
dt = time.perf_counter() - t
t2 = time.perf_counter()
U.convertalis(qdb, db, "/tmp/uc_bench/w_aln", "/tmp/uc_bench/w.m8")
dt2 = time.perf_counter() - t2
rows = sum(1 for _ in open("/tmp/uc_bench/w.m8"))
out["search_5_vs_50"] = {"options": "-c 0.8 (+ search defaults -e 10 --max-seqs 1000)", "wall_s": round(dt, 3), "convertalis_s": round(dt2, 3),
                         "alignments": st["n_gapped_alignments"], "m8_rows": rows, "sw_kernel_ms": round(st["sw_kernel_ms"], 1),
                         "prefilter_kernel_ms": round(st["prefilter_kernel_ms"], 1), "alignments_per_s": round(st["n_gapped_alignments"] / dt)}
print(json.dumps(out, indent=1))
