#!/usr/bin/env python3
"""tools/gaps.py <kernel_trace.csv> [window_ms] — GPU idle gaps inside the LAST bench step of a rocprofv3 kernel trace:
where no kernel is running the host is in the way (synchronisations, allocations, host phases)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 800.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
tend = max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= tend - int(win * 1e6)]
def short(n):
    m = re.search(r"(sw_pk_kernel<[^>]*>|sw_\w+|uc::\w+|radix\w+|scan\w*|transform\w*|fillBuffer\w*|copyBuffer\w*)", n)
    return m.group(1) if m else n[:40]
base = ev[0][0]
cur_e, last_name, gaps, busy = ev[0][1], ev[0][2], [], 0
cur_s = ev[0][0]
for s, e, n in ev[1:]:
    if s > cur_e:
        gaps.append((cur_e - base, s - cur_e, short(last_name), short(n)))
        busy += cur_e - cur_s
        cur_s = s
    if e > cur_e:
        cur_e, last_name = e, n
busy += cur_e - cur_s
tot = sum(g[1] for g in gaps)
print("window %.1f ms, busy %.1f ms, idle %.1f ms in %d gaps" % ((tend - base) / 1e6, busy / 1e6, tot / 1e6, len(gaps)))
for at, ln, a, b in sorted(gaps, key=lambda g: -g[1])[:25]:
    print("  at %7.1f ms  idle %6.2f ms  after %-34s before %s" % (at / 1e6, ln / 1e6, a, b))
