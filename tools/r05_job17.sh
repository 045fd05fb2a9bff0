#!/bin/bash
# r05 GPU job 17: where the device idles during one default-workflow call with configs[3]'s options @ 500 (gaps of the kernel trace by neighbouring kernels)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > /dev/null 2>&1
d=/tmp/prof_gaps; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace -d $d -o out --output-format csv -- python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > $d/run.log 2>&1
python tools/gaps_by_pair.py $d/out_kernel_trace.csv > gpurun_out/job17_gaps_c4_p500.txt; head -45 gpurun_out/job17_gaps_c4_p500.txt
