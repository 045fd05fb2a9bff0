#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
( time python __graft_entry__.py smoke ) > gpurun_out/final_build_smoke.log 2>&1; echo "build+smoke rc=$?"; tail -3 gpurun_out/final_build_smoke.log
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=12 ) > gpurun_out/final_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -22 gpurun_out/final_gpu_tests.log
( time timeout 2700 python bench.py ) > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/final_bench.json; tail -4 gpurun_out/final_bench.err
export UC_ALLOW_SYNTHETIC=1
timeout 1500 python tools/property_campaign.py 6000 480 60 > gpurun_out/final_property_campaign.log 2>&1; tail -3 gpurun_out/final_property_campaign.log
