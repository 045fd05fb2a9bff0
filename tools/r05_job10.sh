#!/bin/bash
# r05 GPU job 10: the whole GPU suite as the driver runs it (with durations), then the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/job10_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -25 gpurun_out/job10_gpu_tests.log
( time timeout 2400 python bench.py ) > gpurun_out/job10_bench.json 2> gpurun_out/job10_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/job10_bench.json; tail -5 gpurun_out/job10_bench.err
