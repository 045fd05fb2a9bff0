#!/usr/bin/env python3
"""tools/shard_time.py — on the GPU box: prefilter time of ONE target shard of N (what each rank of an N-GPU run does)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, unicore_amd as U
from unicore_amd import dist as ucdist
wd = "/tmp/uc_bench/p50"
bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002)
prefix = os.path.join(wd, "db")
lens = bench.read_lens(prefix)
e = U.Engine("-c 0.8", verbosity=1); e.load_db(prefix)
for world in (1, 2, 4, 8):
    tb, te = ucdist.shard_ranges(lens, world)[world // 2]
    e.prefilter(tb, te); e.reset_stats(); e.prefilter(tb, te)
    st = e.stats()
    print("world %d: shard prefilter kernels %.1f ms, hits %d" % (world, st["prefilter_kernel_ms"], e.hits_size()))
