#!/usr/bin/env python3
"""tools/c5_at_size.py [proteomes=500] — BASELINE configs[4]'s chain at its NOMINAL size behind the checks of
tests/test_configs_gpu.py::test_c5_chain_at_40_proteomes: ProstT5 AA -> 3Di encoder over all sequences (~21 min of MFMA
work at 500 proteomes) -> uc_engine_set_db -> cluster step; 3Di states of 200 random sequences == the fp32 restatement, hit lists and
alignment records of 300 random queries == the CPU oracle on the encoder's 3Di track at full database size, TSV invariants.
Run on the GPU box; writes gpurun_out/c5_p<N>_check.json (copied to profiles/r04/)."""
import json, os, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: F401
from oracle import oracle_py as O
import test_configs_gpu as T
prot = int(sys.argv[1]) if len(sys.argv) > 1 else 500
d = tempfile.mkdtemp(prefix="uc_c5_%d_" % prot, dir="/tmp")
t0 = time.time()
res = T.c5_chain_checks(O, d, prot, 200, 300)
res["total_wall_s"] = time.time() - t0
res["checks"] = "all asserts of c5_chain_checks passed"
out = os.path.join(ROOT, "gpurun_out", "c5_p%d_check.json" % prot)
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
