#!/usr/bin/env python3
"""tools/full_parity.py — one-off, on the GPU box: BASELINE configs[1] at FULL size through the CPU oracle (all host
threads, ~20 min) and through the product (uc_cluster + uc_createtsv); compares clust.tsv byte for byte and the stage
counters.  Result recorded in DESIGN.md 5."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, util, unicore_amd as U
from oracle import oracle_py as O
wd = "/tmp/uc_bench/p50"
bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002)
db = os.path.join(wd, "db")
t = time.perf_counter()
st = U.cluster(db, wd + "/gpu_cluster", wd + "/tmp", "-c 0.8", threads=8)
U.createtsv(db, wd + "/gpu_cluster", wd + "/gpu.tsv")
t_gpu = time.perf_counter() - t
odb = O.OracleDb(db)
p = util.oracle_params(O, "-c 0.8")
t = time.perf_counter()
ref = O.cluster(odb, p, threads=len(os.sched_getaffinity(0)), dumps=False)
O.write_tsv(wd + "/cpu.tsv", odb, ref["assign"])
t_cpu = time.perf_counter() - t
g, c = open(wd + "/gpu.tsv", "rb").read(), open(wd + "/cpu.tsv", "rb").read()
print("tsv_identical", g == c, "cpu_wall_s", round(t_cpu, 1), file=sys.stderr, flush=True)
out = {"sequences": odb.n, "tsv_identical": g == c, "tsv_bytes": len(g), "sha256_gpu": hashlib.sha256(g).hexdigest(),
       "sha256_cpu": hashlib.sha256(c).hexdigest(), "gpu_wall_s": round(t_gpu, 2), "cpu_wall_s": round(t_cpu, 1),
       "cpu_threads": len(os.sched_getaffinity(0)),
       "counters_gpu_cpu": {a: [int(st[a]), int(ref["counts"][b])] for a, b in (("n_sim_kmers", "n_sim_kmers"), ("n_kmer_hits", "n_kmer_hits"),
                          ("n_candidates", "n_candidates"), ("n_prefilter_hits", "n_prefilter_hits"), ("n_gapped_alignments", "n_alignments"),
                          ("n_clusters", "n_clusters"), ("cells_fwd", "cells_fwd"), ("cells_rev", "cells_rev"), ("cells_start", "cells_start"))}}
print(json.dumps(out, indent=1))
