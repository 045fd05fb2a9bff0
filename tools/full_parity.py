#!/usr/bin/env python3
"""tools/full_parity.py [single|workflow|both] — on the GPU box: BASELINE configs[1] at FULL size through the CPU oracle (all
host threads) and through the product (uc_cluster + uc_createtsv); compares clust.tsv byte for byte and the stage counters.
  workflow: a bare `-c 0.8` = Foldseek's default workflow (linear-time pre-step + 3-step cascade), oracle uco_cluster_workflow (~4 min)
  single:   `--single-step-clustering`, the plain all-vs-all step the bench times, oracle uco_cluster (~20 min on 256 threads)
Result: profiles/<tag>_full_size_parity.json."""
import hashlib, json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import torch  # noqa: F401
import bench, util, unicore_amd as U
from oracle import oracle_py as O
what = sys.argv[1] if len(sys.argv) > 1 else "both"
wd = "/tmp/uc_bench/p50"
bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002)
db = os.path.join(wd, "db")
odb = O.OracleDb(db)
p = util.oracle_params(O, "-c 0.8")
nthr = len(os.sched_getaffinity(0))
CNT = (("n_sim_kmers", "n_sim_kmers"), ("n_kmer_hits", "n_kmer_hits"), ("n_candidates", "n_candidates"), ("n_prefilter_hits", "n_prefilter_hits"),
       ("n_gapped_alignments", "n_alignments"), ("n_clusters", "n_clusters"), ("cells_fwd", "cells_fwd"), ("cells_rev", "cells_rev"), ("cells_start", "cells_start"))
out = {"sequences": odb.n, "cpu_threads": nthr}
for leg, opts in (("workflow", "-c 0.8"), ("single", "-c 0.8 --single-step-clustering")):
    if what not in (leg, "both"):
        continue
    t = time.perf_counter()
    st = U.cluster(db, wd + "/gpu_cluster", wd + "/tmp", opts, threads=8)
    U.createtsv(db, wd + "/gpu_cluster", wd + "/gpu.tsv")
    t_gpu = time.perf_counter() - t
    t = time.perf_counter()
    if leg == "single":
        ref = O.cluster(odb, p, threads=nthr, dumps=False)
    else:
        ref = O.cluster_workflow(odb, p, O.cascade_thresholds(p, 4.0, 3), linclust_m=20, threads=nthr)
    O.write_tsv(wd + "/cpu.tsv", odb, ref["assign"])
    t_cpu = time.perf_counter() - t
    g, c = open(wd + "/gpu.tsv", "rb").read(), open(wd + "/cpu.tsv", "rb").read()
    print(leg, "tsv_identical", g == c, "cpu_wall_s", round(t_cpu, 1), file=sys.stderr, flush=True)
    out[leg] = {"options": opts, "tsv_identical": g == c, "tsv_bytes": len(g), "sha256_gpu": hashlib.sha256(g).hexdigest(), "sha256_cpu": hashlib.sha256(c).hexdigest(),
                "gpu_wall_s": round(t_gpu, 2), "cpu_wall_s": round(t_cpu, 1),
                "counters_gpu_cpu": {a: [int(st[a]), int(ref["counts"][b])] for a, b in CNT if a in st and b in ref["counts"]}}
    print(json.dumps({leg: out[leg]}), flush=True)
print(json.dumps(out, indent=1))
