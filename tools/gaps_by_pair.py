#!/usr/bin/env python3
"""tools/gaps_by_pair.py <kernel_trace.csv> — GPU idle time of a whole rocprofv3 kernel trace, summed by (kernel before the gap -> kernel after it):
where the host is in the way of the device (synchronisations for counts, allocations, host phases), ranked."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
def short(n):
    m = re.search(r"(sw_pk_kernel|sw_long_kernel|sw_group_kernel|uc::(?:\(anonymous namespace\)::)?\w+|onesweep\w*|radix_sort\w*|scan\w*|lookback\w*|transform\w*|fillBuffer\w*|copyBuffer\w*|histogram\w*)", n)
    return m.group(1).replace("(anonymous namespace)::", "") if m else n[:40]
cur_e, last = ev[0][1], ev[0][2]
by = collections.defaultdict(lambda: [0, 0])
busy_until = ev[0][1]; idle = 0
for s, e, n in ev[1:]:
    if s > cur_e:
        k = (short(last), short(n)); by[k][0] += s - cur_e; by[k][1] += 1; idle += s - cur_e
    if e > cur_e: cur_e, last = e, n
span = ev[-1][1] - ev[0][0]
print("trace span %.1f ms, device idle %.1f ms (%.1f %%) in %d gaps" % (span / 1e6, idle / 1e6, 100.0 * idle / span, sum(v[1] for v in by.values())))
for (a, b), (ns, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:40]:
    print("  %9.1f ms in %6d gaps (%.3f ms each)  after %-36s before %s" % (ns / 1e6, c, ns / 1e6 / c, a, b))
