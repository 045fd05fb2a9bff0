#!/bin/bash
# r05 GPU job 14: two-buffer pipeline of the MODE 7 batches: parity tests that touch the traceback, timing @ 500 and at nominal size
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "traceback or stage_parity or search or random or property or workflow or linclust or set_cover" > gpurun_out/job14_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/job14_tests.log
timeout 900 python -m pytest tests/test_workflow_gpu.py -x -q -m gpu -k "round_hook or c4-200" > gpurun_out/job14_tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/job14_tests2.log
UC_TIMING=1 timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job14_line.json 2> gpurun_out/job14_timing.log; echo "rc=$?"
grep "sw pass mode 7" gpurun_out/job14_timing.log | grep -v "band 0" | cut -c1-200
UC_TIMING=1 timeout 900 python tools/workflow_at_size.py 2000 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job14_line_nominal.json 2> gpurun_out/job14_timing_nominal.log; echo "rc=$?"
tail -c 500 gpurun_out/job14_line_nominal.json
