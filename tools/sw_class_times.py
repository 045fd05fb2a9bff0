#!/usr/bin/env python3
"""per-class / per-mode time of the packed SW kernels from a rocprofv3 kernel-stats CSV (use a UC_STREAMS=1 trace)
usage: sw_class_times.py gpurun_out/prof_<tag>_serial [calls_divisor]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1].rstrip("/") + "/out_kernel_stats.csv")))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(float); bymode = collections.defaultdict(float); byG = collections.defaultdict(float)
for r in rows:
    m = re.search(r"sw_pk_kernel<(\d+), (\d+), (\d+), (\d+)>", r["Name"])
    if not m:
        continue
    G, R, M, NW = map(int, m.groups()); t = int(r["TotalDurationNs"]) / 1e6 / div
    agg[(M, G, R)] += t; bymode[M] += t; byG[G] += t
print("total %.1f ms  by mode %s  by G %s" % (sum(bymode.values()), {k: round(v, 1) for k, v in sorted(bymode.items())}, {k: round(v, 1) for k, v in sorted(byG.items())}))
for M in sorted(bymode):
    print("mode %d: " % M + "  ".join("(%d,%d) %.1f" % (G, R, agg[(M, G, R)]) for (m, G, R) in sorted(agg) if m == M))
