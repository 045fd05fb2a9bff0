#!/bin/bash
# r05 GPU job 3: banded MODE 7 — parity tests that touch the traceback, then per-pass timing at configs[3]'s options @ 500
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "traceback or stage_parity or search or random or property" > gpurun_out/job3_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/job3_tests.log
timeout 600 python -m pytest tests/test_workflow_gpu.py -x -q -m gpu -k "round_hook or c4-200" > gpurun_out/job3_tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/job3_tests2.log
UC_TIMING=1 timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job3_line.json 2> gpurun_out/job3_timing.log; echo "rc=$?"
tail -c 300 gpurun_out/job3_line.json
UC_TB_BAND=24 UC_TIMING=1 timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job3_line_w24.json 2> gpurun_out/job3_timing_w24.log; echo "rc=$?"
