#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export UC_ALLOW_SYNTHETIC=1
o=gpurun_out/ab_tb; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $o/parity.log 2>&1; echo "parity rc=$?"; tail -3 $o/parity.log
timeout 900 python -m pytest tests/test_workflow_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "c4-200 or c4-lite or round_hook" --durations=5 > $o/c4.log 2>&1; echo "c4 rc=$?"; tail -12 $o/c4.log
timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > $o/wf500.json 2> $o/wf500.err; echo "wf rc=$?"; cat $o/wf500.json
timeout 600 python tools/workflow_at_size.py 500 "-c 0.8 --min-seq-id 0.3 -s 7.5" > $o/wf500b.json 2> $o/wf500b.err; echo "wf rc=$?"; cat $o/wf500b.json
