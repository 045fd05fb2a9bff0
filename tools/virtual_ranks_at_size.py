#!/usr/bin/env python3
"""tools/virtual_ranks_at_size.py — BASELINE configs[1] through uc_cluster with 1 / 2 / 4 / 8 ranks and a Q2 x T2 grid on ONE GPU
(UC_VIRTUAL_GPUS=1: several engines per device, the exchange runs through the same code with device copies instead of RCCL):
clust.tsv hash, alignment and cluster counts must not depend on the number of ranks.  Wall times mean nothing here."""
import os, sys, time, hashlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
os.environ["UC_ALLOW_SYNTHETIC"] = "1"
import torch, bench, unicore_amd as U
prefix = bench.gen_db("/tmp/uc_bench/p50_f6000_s1_5eed0002", 50, 6000, 1.0, 0x5EED0002)
ref = None
for n, extra in ((1, ""), (2, ""), (4, ""), (8, ""), (4, " --target-shards 2")):
    os.environ["UC_VIRTUAL_GPUS"] = "1"
    t0 = time.time()
    st = U.cluster(prefix, "/tmp/uc_bench/vg_cluster", "/tmp/uc_bench/tmp", "-c 0.8 --single-step-clustering" + extra, threads=32, num_gpus=n)
    dt = time.time() - t0
    U.createtsv(prefix, "/tmp/uc_bench/vg_cluster", "/tmp/uc_bench/vg.tsv")
    h = hashlib.sha256(open("/tmp/uc_bench/vg.tsv", "rb").read()).hexdigest()[:16]
    if ref is None: ref = h
    print("ranks", n, extra, "wall %.2f" % dt, "aln", st["n_gapped_alignments"], "clusters", st["n_clusters"], "exchange_s %.3f" % st["exchange_seconds"], "bytes", st["exchange_bytes"], "tsv", h, "SAME" if h == ref else "DIFFERENT", flush=True)
