#!/bin/bash
# r05 GPU job 10b: the default bench line (with the new configs.c4 sub-record)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
( time timeout 2700 python bench.py ) > gpurun_out/job10_bench.json 2> gpurun_out/job10_bench.err; echo "bench rc=$?"; tail -c 1200 gpurun_out/job10_bench.json; tail -5 gpurun_out/job10_bench.err
