#!/bin/bash
# tools/prof_serial.sh <tag> [bench args] — kernel trace of one bench step with the class kernels serialized (UC_STREAMS=1):
# per-kernel durations add up to the HIP-event time of the bench line
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
d=gpurun_out/prof_${TAG}_serial
mkdir -p "$d"
UC_STREAMS=1 rocprofv3 --kernel-trace --stats -d "$d" -o out --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs "$@" > "$d/bench.log" 2>&1
echo "rc=$?"; tail -c 300 "$d/bench.log"
