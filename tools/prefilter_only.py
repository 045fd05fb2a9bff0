#!/usr/bin/env python3
"""tools/prefilter_only.py [--config c2] — ONE pass of E1-E4 (index, similar-k-mer match + double-hit filter, ungapped, top-M) and nothing else in the process:
under `rocprofv3 --pmc` every dispatch of the run is a prefilter dispatch, the rocPRIM sorts / scans included (tools/prefilter_traffic_all.sh).  Prints the stage's
algorithmic bytes, kernel time and counts as one JSON line."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import bench
import unicore_amd as U
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--work", default="/tmp/uc_bench")
a = ap.parse_args()
prot, fam, scale, seed, opts, _ = bench.CONFIGS[a.config]
db = bench.gen_db(os.path.join(a.work, "p%d_f%d_s%g_%x" % (prot, fam, scale, seed)), prot, fam, scale, seed)
e = U.Engine(opts, threads=4)
e.load_db(db)
e.prefilter()
st = e.stats()
ab = dict(zip(U.STAGES, st["algorithmic_bytes"]))
print(json.dumps({"config": a.config, "algorithmic_bytes": ab["index"] + ab["kmer"] + ab["ungapped"] + ab["select"], "by_stage": {k: ab[k] for k in ("index", "kmer", "ungapped", "select")},
                  "prefilter_kernel_ms": st["prefilter_kernel_ms"], "n_kmer_hits": st["n_kmer_hits"], "n_candidates": st["n_candidates"], "n_prefilter_hits": st["n_prefilter_hits"]}))
