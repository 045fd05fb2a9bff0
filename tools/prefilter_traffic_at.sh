#!/bin/bash
# tools/prefilter_traffic_at.sh <tag> <bench.py arguments...> — HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes, no trace domains) of the hand-written
# E1-E4 kernels for one bench.py step of any configuration, next to the algorithmic bytes of the same line -> gpurun_out/<tag>_prefilter_traffic.json
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
TAG=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/ptraf_${TAG}_$c; rm -rf $d; mkdir -p $d
  rocprofv3 --pmc $c -d $d -o out --output-format csv -- python bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --no-sub-records > $d/bench.log 2>&1
  echo "$c rc=$?"
done
python - <<PY
import csv, json, collections
tot = collections.defaultdict(float); per = collections.defaultdict(float)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open("gpurun_out/ptraf_${TAG}_%s/out_counter_collection.csv" % c)):
        n = r["Kernel_Name"]
        if "sw_pk_kernel" in n or "sw_group" in n or "sw_long" in n or "rocprim" in n or not n.lstrip("void ").startswith("uc::"):
            continue
        if any(k in n for k in ("plan_", "gate", "finalize", "tb_", "edge_", "sc_", "cells", "aln_", "umark", "uscatter", "pk_flag", "pk_gather", "sm_", "db_pad", "lc_")):
            continue      # gapped-stage / set-cover / layout helpers
        tot[c] += float(r["Counter_Value"]); per[n.split("(")[0][-40:] + ":" + c] += float(r["Counter_Value"])
line = json.loads([l for l in open("gpurun_out/ptraf_${TAG}_FETCH_SIZE/bench.log") if l.startswith("{")][-1])
alg = line["roofline_prefilter"]["algorithmic_bytes_per_step"]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the hand-written E1-E4 kernels of ONE step of: bench.py $*",
       "fetch_size_kib": tot["FETCH_SIZE"], "write_size_kib": tot["WRITE_SIZE"], "bytes_per_step": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0,
       "algorithmic_bytes_per_step": alg, "traffic_over_algorithmic": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / alg,
       "per_kernel_kib": dict(sorted(per.items(), key=lambda x: -x[1])[:16]),
       "note": "KiB units of rocprofv3; gfx950 caveat: FETCH_SIZE counts 1/2 of wide coalesced streams (MI355X_MICROARCH.md), gathers are counted in full; rocPRIM sorts not included"}
json.dump(out, open("gpurun_out/${TAG}_prefilter_traffic.json", "w"), indent=1)
print(json.dumps(out)[:900])
PY
