#!/usr/bin/env python3
"""tools/pass_timeline.py <kernel_trace.csv> [mode] — the launches of one SW pass mode (default 4) out of a rocprofv3 kernel trace, in start order, with their
offsets from the pass's first launch, durations, grids and LDS: how concurrent are the class kernels of a SMALL pass really?  (VERDICT r05 item 4 i.)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ks = []
for r in rows:
    m = re.search(r"sw_pk_kernel<(\d+), (\d+), (\d+), (\d+)>", r["Kernel_Name"])
    if m and int(m.group(3)) == mode:
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(m.group(1)), int(m.group(2)), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                   int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0), int(r.get("LDS_Block_Size", 0) or 0), r.get("Queue_Id", ""), r.get("Stream_Id", "")))
ks.sort()
# split into passes: a gap of > 5 ms between launches starts a new pass
passes, cur = [], []
for k in ks:
    if cur and k[0] - max(x[1] for x in cur) > 5_000_000:
        passes.append(cur); cur = []
    cur.append(k)
if cur: passes.append(cur)
for pi, p in enumerate(passes[-2:]):
    t0 = p[0][0]
    end = max(x[1] for x in p)
    print("pass %d of mode %d: %d launches, span %.2f ms, summed durations %.2f ms" % (pi, mode, len(p), (end - t0) / 1e6, sum(x[1] - x[0] for x in p) / 1e6))
    for s, e, G, R, grid, wg, lds, q, st in p:
        print("  +%7.3f ms  dur %6.3f ms  <%2d,%2d>  workgroups %6d  lds %6d  queue %s stream %s" % ((s - t0) / 1e6, (e - s) / 1e6, G, R, grid // max(wg, 1), lds, q, st))
