#!/usr/bin/env python3
"""tools/cov_stats.py — on the GPU box: how many E-value passers could be rejected from their END positions alone
(coverage upper bound) before the start pass; plus mirror statistics.  Analysis aid, not part of the product."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, unicore_amd as U
wd = "/tmp/uc_bench/p50"
bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002)
prefix = os.path.join(wd, "db")
lens = bench.read_lens(prefix)
e = U.Engine("-c 0.8", verbosity=1)
e.load_db(prefix)
e.prefilter(); cnt, hits = e.hits(); e.align(); al = e.alns()
q = np.repeat(np.arange(len(cnt)), cnt); t = hits["target"]
pe = al["pass_evalue"] == 1
lq = lens[q][pe]; lt = lens[t][pe]
ub_q = (al["qend"][pe] + 1) / lq; ub_t = (al["tend"][pe] + 1) / lt
print("passers", pe.sum(), "accepted", (al["accepted"] == 1).sum())
print("end-bound rejects (cov 0.8, mode 0):", ((ub_q < 0.8) | (ub_t < 0.8)).sum())
print("len ratio rejects (min/max < 0.8):", (np.minimum(lq, lt) / np.maximum(lq, lt) < 0.8).sum())
