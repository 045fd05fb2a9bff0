#!/bin/bash
# tools/filter_ab.sh — A/B of the double-hit filter kernel variants on BASELINE configs[1] (run on the GPU box):
# UC_FILTER_VARIANT 0 = r3 kernel (1024-run tiles, two independent hash positions), 1 = 2048-run tiles, 2 = blocked Bloom, 3 = both
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for v in ${VARIANTS:-0 1 2 3}; do
  UC_FILTER_VARIANT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('variant $v: ms/step %.1f  prefilter kernels %.1f ms  filtered hits %d  aln %d' % (d['ms_per_step'], d['prefilter_kernel_ms_per_step'], d['counts_rank0_per_step']['n_filtered_hits'], d['config']['alignments_per_step']))"
done
