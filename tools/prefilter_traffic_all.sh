#!/bin/bash
# tools/prefilter_traffic_all.sh <tag> [--config c2] — HBM traffic of ONE prefilter pass with EVERYTHING it dispatches: the hand-written E1-E4 kernels AND the rocPRIM
# radix sorts / scans (prefilter_traffic.json left the sorts out).  The process runs the prefilter only (tools/prefilter_only.py), so every
# dispatch counts; FETCH_SIZE and WRITE_SIZE in separate passes, no trace domains.  -> gpurun_out/<tag>_prefilter_traffic_all.json
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
TAG=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/ptall_${TAG}_$c; rm -rf $d; mkdir -p $d
  rocprofv3 --pmc $c -d $d -o out --output-format csv -- python tools/prefilter_only.py "$@" > $d/run.log 2>&1
  echo "$c rc=$?"
done
python - <<PY
import csv, json, collections, re
tot = collections.defaultdict(float); per = collections.defaultdict(float); grp = collections.defaultdict(float)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open("gpurun_out/ptall_${TAG}_%s/out_counter_collection.csv" % c)):
        if r["Counter_Name"] != c: continue
        n = r["Kernel_Name"]
        v = float(r["Counter_Value"])
        tot[c] += v
        m = re.search(r"uc::(\w+)", n)
        k = m.group(1) if m else ("rocprim (radix sort / scan / merge)" if "rocprim" in n else n[:40])
        per[k + ":" + c] += v
        grp[("rocprim" if "rocprim" in n else "hand-written") + ":" + c] += v
line = json.loads([l for l in open("gpurun_out/ptall_${TAG}_FETCH_SIZE/run.log") if l.startswith("{")][-1])
alg = line["algorithmic_bytes"]
b = (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over EVERY dispatch of a process that runs one prefilter pass (tools/prefilter_only.py $*): hand-written E1-E4 kernels + rocPRIM sorts / scans + the database upload's layout kernel",
       "fetch_size_kib": tot["FETCH_SIZE"], "write_size_kib": tot["WRITE_SIZE"], "bytes_per_step": b, "algorithmic_bytes_per_step": alg, "traffic_over_algorithmic": b / alg,
       "groups_kib": dict(grp), "per_kernel_kib": dict(sorted(per.items(), key=lambda x: -x[1])[:24]), "line": line,
       "note": "KiB units of rocprofv3; gfx950 caveat (MI355X_MICROARCH.md): FETCH_SIZE = TCC_EA0_RDREQ x 64 B counts HALF of a wide coalesced 128-B-request stream, gathers in full - not corrected here; working sets below ~256 MiB are partly served by the Infinity Cache"}
json.dump(out, open("gpurun_out/${TAG}_prefilter_traffic_all.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
