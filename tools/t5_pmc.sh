#!/bin/bash
# tools/t5_pmc.sh — PMC counters of the encoder's GEMM kernels (4-layer model, 400 sequences) for the GEMM variants 1 (single-phase 256 tile)
# and 2 (two-phase persistent 256 tile): where do the cycles go?  Counter passes only (no trace domains beside the kernel dispatch records).
# KPAT=<substring of the kernel name> (default t5_gemm; e.g. KPAT=t5_attention VARIANTS=2) selects other kernels of the encoder.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
KPAT=${KPAT:-t5_gemm}
for v in ${VARIANTS:-1 2}; do
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR"; do
    tag=$(echo $set | cut -d' ' -f1)
    d=gpurun_out/t5pmc_v${v}_${tag}
    mkdir -p $d
    UC_T5_GEMM256=$v rocprofv3 --pmc $set -d $d -o out --output-format csv -- python tools/t5_bench.py 4 400 > $d/bench.log 2>&1
    python - <<PY
import csv, collections
f = "$d/out_counter_collection.csv"
tot = collections.defaultdict(float); calls = collections.Counter()
try:
    for r in csv.DictReader(open(f)):
        if "$KPAT" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "$tag": calls[r["Kernel_Name"].split("(")[0][:40]] += 1
    print("variant $v:", dict(calls), {k: "%.4g" % v for k, v in tot.items()})
except Exception as e:
    print("variant $v: no counters (%s)" % e); print(open("$d/bench.log").read()[-600:])
PY
  done
done
