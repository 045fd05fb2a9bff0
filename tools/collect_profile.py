#!/usr/bin/env python3
"""tools/collect_profile.py <tag> — after `tools/profile.sh <tag>` (+ a default `bench.py > gpurun_out/bench_<tag>.json`):
copies the judged summaries from gpurun_out/ into profiles/:
  <tag>_summary.txt       per-kernel table + per-kernel PMC sums (tools/summarize_prof.py)
  <tag>_kernel_stats.csv  rocprofv3 --kernel-trace --stats, verbatim
  <tag>_pmc_sw.txt        PMC sums over all gapped-SW dispatches of the step
  <tag>_bench.json        the default bench line
  sw_traffic.json         FETCH_SIZE / WRITE_SIZE of the SW kernels: what bench.py reports as roofline.traffic"""
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
print("profiles/%s_*: %d SW launches, VALU insts %g" % (tag, len(launches), tot.get("SQ_INSTS_VALU", 0)))
