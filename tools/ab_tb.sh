#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/r04b; mkdir -p $o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/alloc_cost tools/ubench/alloc_cost.hip 2>/dev/null
/tmp/alloc_cost > $o/alloc_cost2.log 2>&1; cat $o/alloc_cost2.log
