#!/bin/bash
# r05 GPU job 1: refactored workflow test (small + c2), nominal c4 round dump, rocprofv3 kernel trace of configs[3]'s options @ 500 (final state)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
timeout 600 python -m pytest tests/test_workflow_gpu.py -x -q -m gpu -k "round_hook or c2" > gpurun_out/job1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/job1_tests.log
timeout 900 python tools/c4_round_fixture.py dump --name c4 > gpurun_out/job1_dump.log 2>&1; echo "dump rc=$?"; tail -2 gpurun_out/job1_dump.log
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
d=gpurun_out/prof_c4wf_p500; mkdir -p $d
timeout 900 rocprofv3 --kernel-trace --stats -d $d -o out --output-format csv -- python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > $d/run.log 2>&1; echo "prof rc=$?"; tail -c 600 $d/run.log
rm -f $d/out_kernel_trace.csv   # (hundreds of MB)
ls -la $d
