#!/usr/bin/env python3
"""tools/variant_check.py — on the GPU box: the full bench database through every execution variant (packed / int32
kernel, mutual-hit sharing on / off): identical alignment records, edge sets and cluster assignments."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, unicore_amd as U
wd = "/tmp/uc_bench/p50"
bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002)
prefix = os.path.join(wd, "db")
ref = None
for opts in ("-c 0.8", "-c 0.8 --sym-dedup 0", "-c 0.8 --sw-kernel i32", "-c 0.8 --sw-kernel i32 --sym-dedup 0"):
    e = U.Engine(opts, verbosity=1); e.load_db(prefix)
    e.prefilter(); e.align()
    al = e.alns(); ed = e.edges(); a = e.setcover(ed)
    key = np.sort(ed[:, 0].astype(np.uint64) << np.uint64(32) | ed[:, 1].astype(np.uint64))
    cur = (al.tobytes(), key.tobytes(), a.tobytes())
    if ref is None: ref = cur
    print(opts, "| records", len(al), "edges", len(ed), "clusters", int((a == np.arange(len(a))).sum()),
          "| identical to default:", cur[0] == ref[0], cur[1] == ref[1], cur[2] == ref[2])
    del e
