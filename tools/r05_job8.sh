#!/bin/bash
# r05 GPU job 8: after the switch clean-up and the plan-wise MODE 7 batches: the fast part of the GPU suite, then per-pass timing at nominal configs[3]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_cli_gpu.py tests/test_golden.py tests/test_multi_gpu.py -x -q -m gpu > gpurun_out/job8_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/job8_tests.log
timeout 900 python -m pytest tests/test_workflow_gpu.py -x -q -m gpu -k "round_hook or c2 or c4-200" > gpurun_out/job8_tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/job8_tests2.log
UC_TIMING=1 timeout 900 python tools/workflow_at_size.py 2000 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job8_line_nominal.json 2> gpurun_out/job8_timing_nominal.log; echo "rc=$?"
tail -c 600 gpurun_out/job8_line_nominal.json
