#!/bin/bash
# final-state evidence of the round, part 1: rocprofv3 passes of the default bench step, the default bench line with its sub-records,
# and the optional length gate (UC-1/L) at configs[1] / configs[2] sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export UC_ALLOW_SYNTHETIC=1
mkdir -p gpurun_out
bash tools/profile.sh r04f > gpurun_out/profile_r04f.log 2>&1; tail -3 gpurun_out/profile_r04f.log
timeout 600 python bench.py > gpurun_out/bench_r04f.json 2> gpurun_out/bench_r04f.err; echo "bench rc=$?"; tail -c 400 gpurun_out/bench_r04f.json
timeout 300 python bench.py --options "-c 0.8 --length-gate 1" --no-sub-records --no-extra-legs --cpu-seconds 8 > gpurun_out/len_gate_c2.json 2> gpurun_out/len_gate_c2.err; echo "len gate c2 rc=$?"
timeout 400 python bench.py --config c3 --options "-c 0.8 --length-gate 1" --steps 1 --warmup 1 --no-cpu-baseline --no-sub-records --no-extra-legs > gpurun_out/len_gate_c3.json 2> gpurun_out/len_gate_c3.err; echo "len gate c3 rc=$?"
python - <<'PY'
import json
for f in ("bench_r04f", "len_gate_c2", "len_gate_c3"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.3g" % d["value"], "ms_per_step %.1f" % d["ms_per_step"], "sw_ms %.1f" % d["roofline"]["kernel_ms_per_step"], "pre_ms %.1f" % d["roofline_prefilter"]["kernel_ms_per_step"], d["config"]["alignments_per_step"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
