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
print("unordered pairs", len(lq), "of directed", len(q))
TABLES = {
    "current": ([64, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 1280, 1536, 1792, 2048], [16] * 6 + [32] * 3 + [64] * 5),
    "R%2 (32/64/128-row steps)": (list(range(32, 385, 32)) + list(range(448, 769, 64)) + list(range(896, 1537, 128)) + [1792, 2048],
                                  [16] * 12 + [32] * 6 + [64] * 6 + [64, 64]),
    "G=8 below 192": ([32, 64, 96, 128, 160, 192, 256, 320, 384, 512, 640, 768, 1024, 1280, 1536, 1792, 2048], [8] * 6 + [16] * 3 + [32] * 3 + [64] * 5),
}
for name, (caps, G) in TABLES.items():
    caps, G = np.array(caps), np.array(G)
    cls = np.searchsorted(caps, lq, side="left")
    tot_u = tot_p = tot_f = 0
    issued = []
    for c in range(len(caps)):
        m = cls == c
        if not m.any(): issued.append(0); continue
        useful = (lq[m] * lt[m]).sum(); padded = (caps[c] * lt[m]).sum(); filled = (caps[c] * (lt[m] + G[c] - 1)).sum()
        tot_u += useful; tot_p += padded; tot_f += filled; issued.append(filled)
    print("%-28s rows-eff %.3f fill-eff %.3f total %.3f  issued-share by class: %s" % (name, tot_u / tot_p, tot_p / tot_f, tot_u / tot_f,
          " ".join("%d:%.2f" % (caps[c], issued[c] / tot_f) for c in range(len(caps)) if issued[c])))
