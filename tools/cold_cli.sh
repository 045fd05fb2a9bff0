#!/bin/bash
# tools/cold_cli.sh — one-shot CLI runs (new process, cold allocations: what an unmodified Unicore does): wall time of
# `bin/foldseek cluster` + `createtsv` at BASELINE configs[1] for several key-buffer sizes (UC_HIT_CAP; empty = default)
export UC_ALLOW_SYNTHETIC=1
DB=/tmp/uc_bench/p50_f6000_s1_5eed0002/db
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
bench.gen_db("/tmp/uc_bench/p50_f6000_s1_5eed0002", 50, 6000, 1.0, 0x5EED0002)
PY
cat $DB $DB.index ${DB}_ss ${DB}_ss.index > /dev/null
for cap in 1000000000 "" 4026531840 1000000000 "" 4026531840; do   # the first run also warms the box up (libraries paged in): ignore it
  for opts in "--single-step-clustering" ""; do
    rm -rf /tmp/uc_bench/cold_out* /tmp/uc_bench/cold_tmp
    s=$(date +%s.%N)
    if [ -n "$cap" ]; then export UC_HIT_CAP=$cap; else unset UC_HIT_CAP; fi
    UC_KEEP_SCRATCH=0 bin/foldseek cluster $DB /tmp/uc_bench/cold_out /tmp/uc_bench/cold_tmp -c 0.8 $opts --threads 32 -v 1 2>/dev/null
    bin/foldseek createtsv $DB $DB /tmp/uc_bench/cold_out /tmp/uc_bench/cold_out.tsv --threads 32 -v 1 2>/dev/null
    e=$(date +%s.%N)
    echo "HIT_CAP=${cap:-default} opts=${opts:-default-workflow} wall $(python -c "print(round($e - $s, 2))") s, tsv lines $(wc -l < /tmp/uc_bench/cold_out.tsv)"
  done
done
