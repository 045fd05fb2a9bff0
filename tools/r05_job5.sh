#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
UC_TIMING=1 timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job5_line.json 2> gpurun_out/job5_timing.log; echo "rc=$?"
UC_TB_NOSTORE=1 UC_TIMING=1 timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job5_line_nostore.json 2> gpurun_out/job5_timing_nostore.log; echo "rc=$?"
UC_STREAMS=1 UC_TIMING=1 timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job5_line_s1.json 2> gpurun_out/job5_timing_s1.log; echo "rc=$?"
