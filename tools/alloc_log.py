#!/usr/bin/env python3
"""tools/alloc_log.py <stderr of a run with UC_ALLOC_LOG=1> — device allocations / frees of 64 MiB and more: how many, how much, how long, by size class."""
import re, sys, collections
a = collections.defaultdict(lambda: [0, 0.0, 0.0]); f = collections.defaultdict(lambda: [0, 0.0, 0.0])
for l in open(sys.argv[1], errors="replace"):
    m = re.search(r"\[alloc\]: (free )?([\d.]+) GiB in ([\d.]+) ms", l)
    if not m: continue
    g, ms = float(m.group(2)), float(m.group(3))
    b = "< 1 GiB" if g < 1 else "1-4 GiB" if g < 4 else "4-16 GiB" if g < 16 else "16-64 GiB" if g < 64 else ">= 64 GiB"
    d = f if m.group(1) else a
    d[b][0] += 1; d[b][1] += g; d[b][2] += ms
for name, d in (("hipMalloc", a), ("hipFree", f)):
    print("%s: %d calls, %.0f GiB, %.1f s" % (name, sum(x[0] for x in d.values()), sum(x[1] for x in d.values()), sum(x[2] for x in d.values()) / 1e3))
    for b in ("< 1 GiB", "1-4 GiB", "4-16 GiB", "16-64 GiB", ">= 64 GiB"):
        if b in d: print("   %-10s %5d calls %8.0f GiB %8.2f s" % (b, d[b][0], d[b][1], d[b][2] / 1e3))
