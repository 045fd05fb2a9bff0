#!/usr/bin/env python3
"""tools/collect_profile.py <tag> — after `tools/profile.sh <tag>` (+ a default `bench.py > gpurun_out/bench_<tag>.json`):
copies the judged summaries from gpurun_out/ into profiles/:
  <tag>_summary.txt       per-kernel table + per-kernel PMC sums (tools/summarize_prof.py)
  <tag>_kernel_stats.csv  rocprofv3 --kernel-trace --stats, verbatim
  <tag>_pmc_sw.txt        PMC sums over all gapped-SW dispatches of the step
  <tag>_bench.json        the default bench line
  <tag>_serial_kernel_stats.csv / _serial_summary.txt   the UC_STREAMS=1 trace: per-kernel durations add up to the HIP-event time
  sw_traffic.json         FETCH_SIZE / WRITE_SIZE of the SW kernels: what bench.py reports as roofline.traffic
  prefilter_traffic_handwritten.json  the same for the hand-written E1-E4 kernels only (roofline_prefilter.traffic_per_step comes from tools/prefilter_traffic_all.sh -> profiles/prefilter_traffic.json, sorts included)"""
import collections, csv, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
prof = os.path.join(ROOT, "profiles")
with open(os.path.join(prof, tag + "_summary.txt"), "w") as o:
    o.write(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), os.path.relpath(base, ROOT)],
                           cwd=ROOT, capture_output=True, text=True, check=True).stdout)
shutil.copy(os.path.join(base, "out_kernel_stats.csv"), os.path.join(prof, tag + "_kernel_stats.csv"))
b = os.path.join(ROOT, "gpurun_out", "bench_%s.json" % tag)
if os.path.exists(b):
    shutil.copy(b, os.path.join(prof, tag + "_bench.json"))
tot, launches = collections.OrderedDict(), set()
for sfx in ("_fetch", "_write", "_sq", "_sq2"):
    f = base + sfx + "/out_counter_collection.csv"
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not ("sw_pk_kernel" in n or "sw_group_kernel" in n or "sw_long" in n):
            continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if sfx == "_fetch":
            launches.add(r["Dispatch_Id"])
with open(os.path.join(prof, tag + "_pmc_sw.txt"), "w") as o:
    o.write("# PMC sums over all gapped-SW kernel dispatches (sw_pk_kernel + sw_group_kernel) of one bench step "
            "(tools/profile.sh %s); SQ_* cycle counters are in quad-cycles\n" % tag)
    for k, v in tot.items():
        o.write("%-24s %g\n" % (k, v))
json.dump({"source": "profiles/%s_pmc_sw.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 0)" % tag,
           "fetch_size_kib": tot.get("FETCH_SIZE"), "write_size_kib": tot.get("WRITE_SIZE"), "sw_launches": len(launches),
           "note": "rocprofv3 FETCH_SIZE/WRITE_SIZE are in KiB; sums over all gapped-SW kernel dispatches of one step. gfx950 caveat "
                   "(MI355X_MICROARCH.md HBM): FETCH_SIZE under-reports wide coalesced streams by 2x; this kernel issues byte/dword loads "
                   "that mostly hit L2/MALL, so the figure is uncalibrated and well below the algorithmic bytes."},
          open(os.path.join(prof, "sw_traffic.json"), "w"), indent=1)
# UC_STREAMS=1 trace
sb = base + "_serial"
if os.path.exists(sb + "/out_kernel_stats.csv"):
    shutil.copy(sb + "/out_kernel_stats.csv", os.path.join(prof, tag + "_serial_kernel_stats.csv"))
    rows = list(csv.DictReader(open(sb + "/out_kernel_stats.csv")))
    sw = [r for r in rows if "sw_pk_kernel" in r["Name"] or "sw_group_kernel" in r["Name"] or "sw_long" in r["Name"]]
    line = [l for l in open(sb + "/bench.log") if l.startswith("{")]
    d = json.loads(line[0]) if line else {}
    with open(os.path.join(prof, tag + "_serial_summary.txt"), "w") as o:
        o.write("# UC_STREAMS=1 (class kernels serialized on the engine stream), bench.py --steps 1 --warmup 0: rocprofv3 --kernel-trace --stats\n")
        o.write("gapped-SW kernels: %d launches, %.3f ms summed, %.4f ms average\n" % (sum(int(r["Calls"]) for r in sw), sum(int(r["TotalDurationNs"]) for r in sw) / 1e6,
                                                                                     sum(int(r["TotalDurationNs"]) for r in sw) / 1e6 / max(1, sum(int(r["Calls"]) for r in sw))))
        if d:
            o.write("bench line of the same run: sw_kernel_ms_per_step %.3f, launches %d, avg_launch_ms %.4f (HIP events on the engine stream)\n"
                    % (d["sw_kernel_ms_per_step"], d["roofline"]["launches"], d["roofline"]["avg_launch_ms"]))
        o.write(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sw_class_times.py"), os.path.relpath(sb, ROOT)], cwd=ROOT, capture_output=True, text=True).stdout)
# prefilter traffic (everything that is not a gapped-SW / set-cover / planner kernel of uc_align.hip)
PRE = ("kmer_extract", "kmer_offsets", "sim_runs", "filter_kernel", "run_range", "run_order", "compact_kernel", "diag_select", "ungapped_kernel", "select_key", "rank_flag",
       "hit_scatter", "hit_count", "expand_kernel", "query_kmer", "distinct_kmer", "rank_offsets", "rank_rec", "position_runs", "query_totals", "position_expand")
pt = {}
for sfx, cname in (("_fetch", "FETCH_SIZE"), ("_write", "WRITE_SIZE")):
    f = base + sfx + "/out_counter_collection.csv"
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname and any(k in r["Kernel_Name"] for k in PRE):
            pt[cname] = pt.get(cname, 0.0) + float(r["Counter_Value"])
if pt:
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the hand-written E1-E4 kernels of one bench step (tools/profile.sh %s); rocPRIM sort/scan kernels not included" % tag,
               "fetch_size_kib": pt.get("FETCH_SIZE"), "write_size_kib": pt.get("WRITE_SIZE"),
               "bytes_per_step": (pt.get("FETCH_SIZE", 0) + pt.get("WRITE_SIZE", 0)) * 1024.0,
               "note": "KiB units of rocprofv3; gfx950 caveat: FETCH_SIZE counts 1/2 of wide coalesced streams (MI355X_MICROARCH.md), gathers are counted in full"},
              open(os.path.join(prof, "prefilter_traffic_handwritten.json"), "w"), indent=1)      # (r06: profiles/prefilter_traffic.json - what bench.py reads - is the ALL-dispatch figure of tools/prefilter_traffic_all.sh, rocPRIM sorts included)
for extra in ("c3", "c4l", "c5"):
    b2 = os.path.join(ROOT, "gpurun_out", "bench_%s_%s.json" % (extra, tag))
    if os.path.exists(b2) and os.path.getsize(b2):
        shutil.copy(b2, os.path.join(prof, "%s_bench_%s.json" % (tag, extra)))
print("profiles/%s_*: %d SW launches, VALU insts %g" % (tag, len(launches), tot.get("SQ_INSTS_VALU", 0)))
