#!/bin/bash
# r05 GPU job 2: per-pass timing of configs[3]'s options @ 500 (which SW mode costs what per cell; is MODE 7 bound by its byte stores?)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
UC_TIMING=1 timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job2_line.json 2> gpurun_out/job2_timing.log; echo "rc=$?"
grep -c "sw pass" gpurun_out/job2_timing.log; tail -c 400 gpurun_out/job2_line.json
