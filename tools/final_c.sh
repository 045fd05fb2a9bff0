#!/bin/bash
# final-state evidence of the round, part 2: build() + smoke() on the box, then the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export UC_ALLOW_SYNTHETIC=1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/build_smoke.log 2>&1; tail -2 gpurun_out/build_smoke.log
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -16 gpurun_out/gpu_tests_final.log
