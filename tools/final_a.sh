#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export UC_ALLOW_SYNTHETIC=1
bash tools/profile.sh r04 > gpurun_out/profile_r04.log 2>&1; tail -3 gpurun_out/profile_r04.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/build_smoke.log 2>&1; tail -2 gpurun_out/build_smoke.log
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -16 gpurun_out/gpu_tests_final.log
