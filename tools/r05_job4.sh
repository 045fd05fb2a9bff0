#!/bin/bash
# r05 GPU job 4: per-pass timing at configs[3]'s options @ 500 after a change to MODE 7; the band test
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "traceback" > gpurun_out/job4_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/job4_tests.log
UC_TIMING=1 timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job4_line.json 2> gpurun_out/job4_timing.log; echo "rc=$?"
tail -c 300 gpurun_out/job4_line.json
