#!/bin/bash
# r05 GPU job 16: sim_runs with the last DFS level in one step (leaf groups): prefilter parity, then timing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_cli_gpu.py tests/test_golden.py -x -q -m gpu > gpurun_out/job16_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/job16_tests.log
timeout 900 python -m pytest tests/test_workflow_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "round_hook or c2 or c4-lite" > gpurun_out/job16_tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/job16_tests2.log
python bench.py --no-sub-records --no-extra-legs --no-cpu-baseline > gpurun_out/job16_bench_c2.json 2> /dev/null; python -c "
import json; b=json.loads(open('gpurun_out/job16_bench_c2.json').read().strip().splitlines()[-1]); print('c2: ms/step %.1f prefilter kernels %.1f ms sw %.1f' % (b['ms_per_step'], b['roofline_prefilter']['kernel_ms_per_step'], b['roofline']['kernel_ms_per_step']))"
UC_TIMING=1 timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > gpurun_out/job16_line.json 2> gpurun_out/job16_timing.log; echo "rc=$?"
tail -c 420 gpurun_out/job16_line.json; grep "hits  " gpurun_out/job16_timing.log
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
d=gpurun_out/prof_r05_c4wf_p500_b; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats -d $d -o out --output-format csv -- python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > $d/run.log 2>&1; rm -f $d/out_kernel_trace.csv
grep -i "sim_runs\|filter_kernel\|diag_select" $d/out_kernel_stats.csv | cut -c1-60,150-260
