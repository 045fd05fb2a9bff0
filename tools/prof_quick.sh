#!/bin/bash
# tools/prof_quick.sh <tag> — kernel-trace only (one bench step), for quick per-kernel timing on the GPU box
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
d=gpurun_out/prof_${TAG}
mkdir -p "$d"
rocprofv3 --kernel-trace --stats -d "$d" -o out --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs "$@" > "$d/bench.log" 2>&1
echo "rc=$?"; tail -c 300 "$d/bench.log"
