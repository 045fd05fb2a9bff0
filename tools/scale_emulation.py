#!/usr/bin/env python3
"""tools/scale_emulation.py — on the 1-GPU box: what ONE rank of an N-GPU run does, timed on one GPU (no communication):
prefilter of its (query group, target shard) cell of unicore_amd.dist.grid_ranges, device merge + ownership filter of the full union, gapped stage on its share of the
pairs.  An estimate of the per-rank critical path for DESIGN.md 6, not a measurement of an N-GPU run."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, unicore_amd as U
from unicore_amd import dist as ucdist
wd = "/tmp/uc_bench/p50"
bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002)
prefix = os.path.join(wd, "db")
lens = bench.read_lens(prefix)
e = U.Engine("-c 0.8", verbosity=1); e.load_db(prefix)
dev = torch.device("cuda", 0)
def sync(): torch.cuda.synchronize()
for world in (1, 2, 4, 8):
    shards = ucdist.grid_ranges(lens, world)
    # the union every rank receives: all shards' lists (built once per N, outside the timing)
    parts = []
    for tb, te, qb, qe in shards:
        e.prefilter(tb, te, qb, qe)
        buf, n = ucdist._export_hits_tensor(e, dev)
        parts.append(buf[:, :n].clone())
    allh = torch.cat(parts, dim=1).contiguous(); sync()
    rank = world // 2
    tb, te, qb, qe = shards[rank]
    e.prefilter(tb, te, qb, qe)              # warm
    t0 = time.perf_counter(); e.prefilter(tb, te, qb, qe); t_pre = time.perf_counter() - t0
    ntot = int(allh.shape[1])
    ptrs = [allh[i].data_ptr() for i in range(4)]
    e.hits_import_dev(ntot, *ptrs, rank, world); e.align()      # warm
    t0 = time.perf_counter(); kept = e.hits_import_dev(ntot, *ptrs, rank, world); t_imp = time.perf_counter() - t0
    e.reset_stats(); t0 = time.perf_counter(); e.align(); t_aln = time.perf_counter() - t0
    swk = e.stats()["sw_kernel_ms"]
    ed = e.edges()
    print("N=%d (Q%d x T%d) rank %d: prefilter(cell) %.0f ms | merge+ownership %.0f ms | gapped stage on %d of %d pairs %.0f ms (SW kernels %.0f ms) | sum %.0f ms (+ exchange, + set cover ~9 ms on rank 0)"
          % ((world,) + ucdist.grid_shape(lens, world) + (rank, t_pre * 1e3, t_imp * 1e3, kept, 14107485, t_aln * 1e3, swk, (t_pre + t_imp + t_aln) * 1e3)))
