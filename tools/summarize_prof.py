#!/usr/bin/env python3
"""Summarize rocprofv3 CSV output (tools/profile.sh) into a compact, commit-able text file.
usage: summarize_prof.py gpurun_out/prof_<tag> > profiles/<tag>_summary.txt"""
import csv, re, sys, os, collections

def short(name):
    name = re.sub(r"\(.*", "", name)                   # drop argument lists
    m = re.search(r"(uc::\w+(<[^>]*>)?)", name)
    if m: return m.group(1)
    m = re.search(r"(radix_sort_\w+|onesweep\w*|scan\w*|lookback\w*|transform\w*|histogram\w*)", name)
    return "rocprim::" + (m.group(1) if m else name[-60:])

def main():
    base = sys.argv[1].rstrip("/")
    ks = os.path.join(base, "out_kernel_stats.csv")
    rows = list(csv.DictReader(open(ks)))
    agg = collections.OrderedDict()
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0])
        a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print("# rocprofv3 --kernel-trace --stats (%s)" % base)
    print("%-46s %7s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-46s %7d %12.3f %12.4f %6.2f%%" % (k, c, ns / 1e6, ns / 1e6 / c, 100.0 * ns / tot))
    print("%-46s %7s %12.3f" % ("TOTAL", "", tot / 1e6))
    sw = [(k, v) for k, v in agg.items() if "sw_group_kernel" in k or "sw_pk_kernel" in k or "sw_long" in k]
    c = sum(v[0] for _, v in sw); ns = sum(v[1] for _, v in sw)
    if c: print("\nsw kernels (all classes, all passes): %d launches, %.3f ms total, %.4f ms avg" % (c, ns / 1e6, ns / 1e6 / c))
    for sfx, label in (("_fetch", "FETCH_SIZE"), ("_write", "WRITE_SIZE"), ("_sq", None), ("_sq2", None)):
        f = os.path.join(base + sfx, "out_counter_collection.csv")
        if not os.path.exists(f): continue
        cagg = collections.defaultdict(lambda: collections.defaultdict(float))
        ncall = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            cagg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        print("\n# rocprofv3 --pmc (%s) — summed over dispatches; FETCH_SIZE/WRITE_SIZE in KiB-units of rocprof (x1024 B)" % (base + sfx))
        names = sorted({n for v in cagg.values() for n in v})
        print("%-46s " % "kernel" + " ".join("%18s" % n for n in names))
        def key(kv): return -sum(kv[1].values())
        for k, v in sorted(cagg.items(), key=key)[:24]:
            print("%-46s " % k + " ".join("%18.0f" % v.get(n, 0) for n in names))
        swv = collections.defaultdict(float)
        for k, v in cagg.items():
            if "sw_group_kernel" in k or "sw_pk_kernel" in k:
                for n, x in v.items(): swv[n] += x
        if swv: print("%-46s " % "SUM sw kernels" + " ".join("%18.0f" % swv.get(n, 0) for n in names))

if __name__ == "__main__":
    main()
