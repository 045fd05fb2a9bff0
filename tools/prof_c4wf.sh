#!/bin/bash
# tools/prof_c4wf.sh <tag> [proteomes=500] — rocprofv3 evidence for ONE default-workflow call with BASELINE configs[3]'s options
# ("-c 0.8 --min-seq-id 0.3 -s 7.5", tools/workflow_at_size.py): kernel trace + stats, then PMC passes (separate runs, no trace domains) with the
# counters summed per SW pass mode -> gpurun_out/prof_<tag>/{out_kernel_stats.csv, pmc_by_mode.txt}
set -u
TAG=$1; P=${2:-500}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
export UC_ALLOW_SYNTHETIC=1
OPTS="-c 0.8 --min-seq-id 0.3 -s 7.5"
d=gpurun_out/prof_$TAG; rm -rf $d; mkdir -p $d
python tools/workflow_at_size.py $P "$OPTS" > $d/warm.log 2>&1      # database generation + a warm call outside the profiler
rocprofv3 --kernel-trace --stats -d $d -o out --output-format csv -- python tools/workflow_at_size.py $P "$OPTS" > $d/run.log 2>&1; echo "trace rc=$?"
rm -f $d/out_kernel_trace.csv
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  dd=$d/pmc_$tag; mkdir -p $dd
  rocprofv3 --pmc $set -d $dd -o out --output-format csv -- python tools/workflow_at_size.py $P "$OPTS" > $dd/run.log 2>&1; echo "pmc $tag rc=$?"
done
python - <<PY
import csv, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.Counter()
for f in glob.glob("$d/pmc_*/out_counter_collection.csv"):
    first = None
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"sw_pk_kernel<(\d+), (\d+), (\d+)", n)
        key = ("packed SW, mode %s" % m.group(3)) if m else ("int32 SW (sw_group / sw_long)" if ("sw_group" in n or "sw_long" in n) else ("tb_walk_kernel" if "tb_walk" in n else None))
        if not key: continue
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if first is None: first = r["Counter_Name"]
        if r["Counter_Name"] == first: disp[(key, f)] += 1
names = sorted({c for v in agg.values() for c in v})
with open("$d/pmc_by_mode.txt", "w") as o:
    o.write("# rocprofv3 --pmc, separate passes over: tools/workflow_at_size.py $P '$OPTS' (one default-workflow call); counters summed over the dispatches of each SW pass mode\n")
    o.write("# SQ_* in quad-cycles / wave instructions; FETCH_SIZE / WRITE_SIZE in KiB (gfx950: FETCH_SIZE counts 1/2 of wide coalesced streams)\n")
    o.write("%-32s " % "kernels" + " ".join("%20s" % n for n in names) + "\n")
    for k in sorted(agg):
        o.write("%-32s " % k + " ".join("%20.4g" % agg[k].get(n, 0) for n in names) + "\n")
print(open("$d/pmc_by_mode.txt").read())
PY
ls $d
