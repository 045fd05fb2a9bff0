#!/bin/bash
# tools/cold_stamps.sh — where a one-shot `foldseek cluster` process spends its time (UC_TIMING=1 stamps relative to library load),
# BASELINE configs[1], page cache warm; first run = box warm-up, read the later ones.  Run on the GPU box.
export UC_ALLOW_SYNTHETIC=1 UC_TIMING=1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
DB=/tmp/uc_bench/p50_f6000_s1_5eed0002/db
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
bench.gen_db("/tmp/uc_bench/p50_f6000_s1_5eed0002", 50, 6000, 1.0, 0x5EED0002)
PY
cat $DB $DB.index ${DB}_ss ${DB}_ss.index > /dev/null
for rep in 1 2 3; do
  for opts in "--single-step-clustering" ""; do
    rm -rf /tmp/uc_bench/cold_out* /tmp/uc_bench/cold_tmp
    echo "== run $rep opts='${opts:-default workflow}'"
    s=$(date +%s.%N)
    bin/foldseek cluster $DB /tmp/uc_bench/cold_out /tmp/uc_bench/cold_tmp -c 0.8 $opts --threads 32 -v 1 2>&1 | grep -v Warning
    m=$(date +%s.%N)
    bin/foldseek createtsv $DB $DB /tmp/uc_bench/cold_out /tmp/uc_bench/cold_out.tsv --threads 32 -v 1 2>/dev/null
    e=$(date +%s.%N)
    python3 -c "print('   cluster process %.3f s, createtsv process %.3f s, total %.3f s' % ($m - $s, $e - $m, $e - $s))"
  done
done
