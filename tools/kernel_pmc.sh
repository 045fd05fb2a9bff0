#!/bin/bash
# tools/kernel_pmc.sh <kernel-name substring> <bench.py arguments...> — PMC counters (separate passes, no trace domains) summed over the dispatches of the
# kernels whose name contains the substring, for one bench.py run.  Example: tools/kernel_pmc.sh sim_runs --config c4-lite --steps 1 --warmup 0
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
KPAT=$1; shift
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES_LT_64 GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  d=gpurun_out/kpmc_${KPAT}_${tag}
  rm -rf $d; mkdir -p $d
  rocprofv3 --pmc $set -d $d -o out --output-format csv -- python bench.py "$@" --no-cpu-baseline --no-extra-legs --no-sub-records > $d/bench.log 2>&1
  python - <<PY
import csv, collections, glob
tot = collections.defaultdict(float); n = 0
for f in glob.glob("$d/**/out_counter_collection.csv", recursive=True) + glob.glob("$d/out_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$KPAT" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n += r["Counter_Name"] == "$tag"
print("$KPAT:", n, "dispatches", {k: "%.4g" % v for k, v in tot.items()})
PY
done
