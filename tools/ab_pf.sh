#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export UC_ALLOW_SYNTHETIC=1
o=gpurun_out/ab_pf; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ungapped or pipeline_stage_parity or chunked_prefilter or wide_and_compact or random_option_sets or cluster_end_to_end or degenerate" > $o/tests.log 2>&1; echo "tests rc=$?" ; tail -3 $o/tests.log
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats -d $o/prof -o out --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/prof_bench.json 2> $o/prof_bench.err
python - <<P
import csv,glob,re,json
d=json.loads(open("$o/prof_bench.json").read().strip().splitlines()[-1])
print("ms/step %.1f"%d["ms_per_step"], "prefilter_kernel_ms %.2f"%d["prefilter_kernel_ms_per_step"], "stages", d["stages_s_per_step"], "cand", d["counts_rank0_per_step"]["n_candidates"], "clusters", d["config"]["clusters"])
f=glob.glob("$o/prof/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "sw_pk" in n or "rocprim" in n: continue
    if int(r["TotalDurationNs"])>4e6: print("%9.3f ms/step %4s calls  %s"%(int(r["TotalDurationNs"])/1e6/4,r["Calls"],re.sub(r"\(.*","",n)[:90]))
P
timeout 900 python bench.py --config c4 --workflow plain --proteomes 100 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/c4plain100.json 2> $o/c4plain100.err; echo "c4 plain rc=$?"
python - <<P
import json
d=json.loads(open("$o/c4plain100.json").read().strip().splitlines()[-1])
print("c4 plain@100: ms/step %.1f"%d["ms_per_step"], "prefilter_kernel_ms %.1f"%d["prefilter_kernel_ms_per_step"], "sw %.1f"%d["sw_kernel_ms_per_step"], d["stages_s_per_step"])
P
