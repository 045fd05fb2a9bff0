#!/usr/bin/env python3
"""tools/overlap_trace.py <kernel_trace.csv> — reads a `rocprofv3 --kernel-trace --output-format csv` trace of tools/overlap_probe.py and says, per prefilter
kernel, how long its launches take when NO gapped-stage (SW) kernel is on the device and how long when one is; plus the union / intersection of the two
stages' busy intervals.  The evidence behind DESIGN.md's "co-scheduling the two stages does not pay" (VERDICT r05 item 1)."""
import csv, json, sys
from collections import defaultdict

SW = ("sw_pk_kernel", "sw_group_kernel", "sw_long_kernel")
PRE = ("sim_runs", "filter_kernel", "kmer_", "position_", "rank_rec", "compact_kernel", "diag_select", "diag_long", "ungapped_kernel", "select_key", "rank_flag",
       "hit_scatter", "distinct_kmer", "query_kmer", "run_range", "run_order", "hit_count", "query_totals", "expand_kernel")


def short(n):
    n = n.split("(")[0]
    return n.split("::")[-1].split("<")[0] if "rocprim" not in n else "rocprim"


def union(iv):
    iv = sorted(iv); out = []
    for a, b in iv:
        if out and a <= out[-1][1]: out[-1][1] = max(out[-1][1], b)
        else: out.append([a, b])
    return out


def inter_len(u, v):
    i = j = 0; t = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if a < b: t += b - a
        if u[i][1] < v[j][1]: i += 1
        else: j += 1
    return t


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    sw = [(a, b) for n, a, b in ks if any(s in n for s in SW)]
    pre = [(n, a, b) for n, a, b in ks if any(s in n for s in PRE) and not any(s in n for s in SW)]
    usw = union(sw)
    import bisect
    starts = [u[0] for u in usw]
    def sw_overlap(a, b):
        k = bisect.bisect_right(starts, b) - 1
        t = 0
        while k >= 0 and usw[k][1] > a:
            t += max(0, min(b, usw[k][1]) - max(a, usw[k][0])); k -= 1
        return t
    per = defaultdict(lambda: {"alone": [], "with_sw": []})
    for n, a, b in pre:
        per[short(n)]["with_sw" if sw_overlap(a, b) > 0.5 * (b - a) else "alone"].append(b - a)
    out = {"kernels": {}}
    for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1]["alone"]) - sum(kv[1]["with_sw"])):
        al, ws = d["alone"], d["with_sw"]
        out["kernels"][n] = {"launches_alone": len(al), "avg_ms_alone": sum(al) / max(len(al), 1) / 1e6, "launches_under_sw": len(ws),
                             "avg_ms_under_sw": sum(ws) / max(len(ws), 1) / 1e6,
                             "stretch": (sum(ws) / len(ws)) / (sum(al) / len(al)) if al and ws else None}
    upre = union([(a, b) for _, a, b in pre])
    out["busy_ms"] = {"sw_union": sum(b - a for a, b in usw) / 1e6, "prefilter_union": sum(b - a for a, b in upre) / 1e6, "both_at_once": inter_len(usw, upre) / 1e6}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
