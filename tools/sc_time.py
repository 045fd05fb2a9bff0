import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench, unicore_amd as U
wd = "/tmp/uc_bench/p50"
bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002)
e = U.Engine("-c 0.8", verbosity=1); e.load_db(os.path.join(wd, "db"))
e.prefilter(); e.align(); ed = e.edges(); n = e.n
for _ in range(3):
    t = time.perf_counter(); a = U.setcover(n, ed); print("setcover %.1f ms, %d edges, %d clusters" % ((time.perf_counter() - t) * 1e3, len(ed), (a == np.arange(n)).sum()))
