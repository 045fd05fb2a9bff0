#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
UC_TIMING=1 python tools/workflow_at_size.py 50 "-c 0.8 --single-step-clustering" 0x5EED0002 > gpurun_out/job13_c2.json 2> gpurun_out/job13_c2_timing.log
UC_TIMING=1 python tools/workflow_at_size.py 50 "-c 0.8 --single-step-clustering" 0x5EED0002 > gpurun_out/job13_c2.json 2> gpurun_out/job13_c2_timing.log
grep "sw pass" gpurun_out/job13_c2_timing.log
UC_TIMING=1 python tools/workflow_at_size.py 500 "-c 0.8 --single-step-clustering" 0x5EED0003 > gpurun_out/job13_c3.json 2> gpurun_out/job13_c3_timing.log
grep "sw pass" gpurun_out/job13_c3_timing.log
