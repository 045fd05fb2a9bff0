#!/usr/bin/env python3
"""tools/workflow_probe.py — where the default workflow (pre-step + 3-step cascade) of uc_cluster spends its wall time at C2"""
import os, sys, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import torch  # noqa
import unicore_amd as U
import bench
prefix = bench.gen_db("/tmp/uc_bench/p50_f6000_s1_5eed0002", 50, 6000, 1.0, 0x5EED0002)
for opts in ("-c 0.8", "-c 0.8 --linclust 0", "-c 0.8 --cluster-steps 1", "-c 0.8 --single-step-clustering"):
    for rep in range(2):
        t0 = time.time()
        st = U.cluster(prefix, "/tmp/uc_bench/wp_cluster", "/tmp/uc_bench/tmp", opts, threads=int(sys.argv[1]) if len(sys.argv) > 1 else 64, verbosity=1)
        dt = time.time() - t0
    print(opts, "wall %.3f" % dt, "aln", st["n_gapped_alignments"], "clusters", st["n_clusters"], {k: round(v, 3) for k, v in zip(U.STAGES, st["stage_seconds"])},
          "sw_ms %.0f pre_ms %.0f" % (st["sw_kernel_ms"], st["prefilter_kernel_ms"]))
