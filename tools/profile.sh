#!/bin/bash
# tools/profile.sh <tag> [bench args...] — rocprofv3 evidence for one bench.py configuration (run on the GPU box).
#   pass 1: --kernel-trace --stats          -> gpurun_out/prof_<tag>/kt_*            (per-kernel time)
#   pass 1b: the same with UC_STREAMS=1      -> gpurun_out/prof_<tag>_serial/        (no stream overlap: durations add up)
#   pass 2: --pmc FETCH_SIZE                 -> gpurun_out/prof_<tag>_fetch/         (HBM read side, separate pass)
#   pass 3: --pmc WRITE_SIZE                 -> gpurun_out/prof_<tag>_write/
#   pass 4: SQ counters                      -> gpurun_out/prof_<tag>_sq/
# Counter passes never combine --pmc with trace domains other than the implicit kernel dispatch records.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs $*"
run() {  # dir, rocprof flags...
  local d=gpurun_out/prof_${TAG}$1; shift
  mkdir -p "$d"
  rocprofv3 "$@" -d "$d" -o out --output-format csv -- python bench.py $ARGS > "$d/bench.log" 2>&1
  echo "== $d: rc=$?"; tail -c 600 "$d/bench.log" | tail -2
}
run "" --kernel-trace --stats
# the same with the class kernels serialized on ONE stream: per-kernel durations then add up to the HIP-event time of
# the bench line (with 8 streams they overlap and sum to more than the step)
UC_STREAMS=1 run _serial --kernel-trace --stats
run _fetch --pmc FETCH_SIZE
run _write --pmc WRITE_SIZE
run _sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
run _sq2 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
find gpurun_out -name '*.csv' | head -30
du -sh gpurun_out
