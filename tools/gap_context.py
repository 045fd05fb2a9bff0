#!/usr/bin/env python3
"""tools/gap_context.py <kernel_trace.csv> <name-before> <name-after> [n] — the kernel records around the first n idle gaps between a kernel whose name contains
<name-before> and the next kernel to start, whose name contains <name-after> (what sits between two MODE 7 batches?)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
a, b = sys.argv[2], sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Queue_Id", ""), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in rows)
cur_e, last_i, shown = ev[0][1], 0, 0
for i in range(1, len(ev)):
    s, e = ev[i][0], ev[i][1]
    if s > cur_e and a in ev[last_i][2] and b in ev[i][2] and s - cur_e > 2_000_000:
        t0 = ev[max(0, i - 4)][0]
        print("gap of %.2f ms:" % ((s - cur_e) / 1e6))
        for j in range(max(0, i - 4), min(len(ev), i + 5)):
            print("   +%9.3f .. +%9.3f ms  queue %s  workgroups %7d  %s" % ((ev[j][0] - t0) / 1e6, (ev[j][1] - t0) / 1e6, ev[j][3], ev[j][4], ev[j][2]))
        shown += 1
        if shown >= n: break
    if e > cur_e:
        cur_e, last_i = e, i
