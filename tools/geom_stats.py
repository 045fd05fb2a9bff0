#!/usr/bin/env python3
"""tools/geom_stats.py — on the GPU box: where the issued DP work of the forward pass goes (useful cells vs row
padding vs pipeline fill/drain vs A/B pairing), per systolic class.  Analysis aid, not part of the product."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, unicore_amd as U
wd = "/tmp/uc_bench/p50"
bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002)
prefix = os.path.join(wd, "db")
lens = bench.read_lens(prefix).astype(np.int64)
e = U.Engine("-c 0.8", verbosity=1)
e.load_db(prefix)
e.prefilter(); cnt, hits = e.hits()
q = np.repeat(np.arange(len(cnt)), cnt).astype(np.int64); t = hits["target"].astype(np.int64)
lo, hi = np.minimum(q, t), np.maximum(q, t)
key = lo * (1 << 24) + hi
_, first = np.unique(key, return_index=True)          # one representative per unordered pair
a, b = q[first], t[first]
la, lb = lens[a], lens[b]
lq, lt = np.minimum(la, lb), np.maximum(la, lb)       # shorter sequence is the query
caps = np.array([64, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 1280, 1536, 1792, 2048])
G = np.array([16, 16, 16, 16, 16, 16, 32, 32, 32, 64, 64, 64, 64, 64])
cls = np.searchsorted(caps, lq, side="left")
print("unordered pairs", len(lq), "of directed", len(q))
tot_u = tot_p = tot_f = 0
for c in range(len(caps)):
    m = cls == c
    if not m.any(): continue
    useful = (lq[m] * lt[m]).sum(); padded = (caps[c] * lt[m]).sum(); filled = (caps[c] * (lt[m] + G[c] - 1)).sum()
    tot_u += useful; tot_p += padded; tot_f += filled
    print("class cap %4d G %2d: pairs %8d  rows-eff %.3f  fill-eff %.3f  total %.3f  share-of-issued %.3f" % (caps[c], G[c], m.sum(), useful / padded, padded / filled, useful / filled, 0))
print("ALL: rows-eff %.3f fill-eff %.3f total %.3f" % (tot_u / tot_p, tot_p / tot_f, tot_u / tot_f))
