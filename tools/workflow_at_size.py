#!/usr/bin/env python3
"""tools/workflow_at_size.py <proteomes> "<options>" [seed] — one uc_cluster call from disk (the call of cluster.rs:45-49) at a
given size, `-v 3` + UC_TIMING round stamps on stderr, one JSON line with wall, stage seconds and counters on stdout."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
os.environ.setdefault("UC_TIMING", "1")
import torch  # noqa: F401
import unicore_amd as U
import bench
prot, opts = int(sys.argv[1]), sys.argv[2]
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x5EED0004
wd = "/tmp/uc_bench/p%d_f6000_s1_%x" % (prot, seed)
t0 = time.time()
prefix = bench.gen_db(wd, prot, 6000, 1.0, seed)
t_gen = time.time() - t0
t0 = time.time()
st = U.cluster(prefix, wd + "/wf_cluster", wd + "/tmp", opts, threads=bench.usable_cores()[0], verbosity=3)
dt = time.time() - t0
print(json.dumps({"proteomes": prot, "options": opts, "gen_s": round(t_gen, 1), "wall_s": round(dt, 2), "sequences": st["n_seqs"], "residues": st["n_residues"],
                  "alignments": st["n_gapped_alignments"], "clusters": st["n_clusters"], "sw_kernel_ms": st["sw_kernel_ms"], "prefilter_kernel_ms": st["prefilter_kernel_ms"],
                  "stages_s": {k: round(v, 2) for k, v in zip(U.STAGES, st["stage_seconds"])},
                  "counts": {k: st[k] for k in ("n_sim_kmers", "n_kmer_hits", "n_candidates", "n_prefilter_hits")}}), flush=True)
