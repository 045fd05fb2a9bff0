#!/bin/bash
# r05 GPU job 15: does the number of hardware queues behind the 8 HIP streams matter for the many-small-launch passes?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
export UC_ALLOW_SYNTHETIC=1
python tools/workflow_at_size.py 50 "-c 0.8 --single-step-clustering" 0x5EED0002 > /dev/null 2>&1
for q in default 2 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for rep in 1 2; do
    UC_TIMING=1 python tools/workflow_at_size.py 50 "-c 0.8 --single-step-clustering" 0x5EED0002 2> gpurun_out/job15_c2_q$q.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 queues $q rep $rep: sw_kernel_ms %.1f prefilter %.1f wall %.2f' % (d['sw_kernel_ms'], d['prefilter_kernel_ms'], d['wall_s']))"
  done
  grep "sw pass" gpurun_out/job15_c2_q$q.log | awk '{print "   ", $5, $6, $12, $13}'
done
for q in default 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  UC_TIMING=1 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" 2> gpurun_out/job15_c4_q$q.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4-500 queues $q: sw_kernel_ms %.1f prefilter %.1f wall %.2f' % (d['sw_kernel_ms'], d['prefilter_kernel_ms'], d['wall_s']))"
done
