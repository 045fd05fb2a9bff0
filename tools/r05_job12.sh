#!/bin/bash
# r05 GPU job 12: attention over HG = 4 heads per workgroup: encoder parity tests, then throughput (8-block probe, 24-block model)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
timeout 1200 python -m pytest tests/test_t5.py -x -q -m gpu > gpurun_out/job12_tests.log 2>&1; echo "t5 tests rc=$?"; tail -4 gpurun_out/job12_tests.log
python tools/t5_bench.py 8 1200 > gpurun_out/job12_t5_bench.log 2>&1; tail -1 gpurun_out/job12_t5_bench.log
python tools/t5_bench.py 8 1200 >> gpurun_out/job12_t5_bench.log 2>&1; tail -1 gpurun_out/job12_t5_bench.log
python tools/t5_bench.py 24 2000 >> gpurun_out/job12_t5_bench.log 2>&1; tail -1 gpurun_out/job12_t5_bench.log
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
d=gpurun_out/prof_r05_t5; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats -d $d -o out --output-format csv -- python tools/t5_bench.py 8 1200 > $d/run.log 2>&1; rm -f $d/out_kernel_trace.csv
head -8 $d/out_kernel_stats.csv | cut -c1-150
