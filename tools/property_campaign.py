#!/usr/bin/env python3
"""tools/property_campaign.py <first_seed> <n_seeds> — the property test of tests/test_gpu_parity.py (random small databases x
random option strings against the oracle: hit lists, every alignment field, set cover) for seeds beyond the 16 of the suite;
every third seed also forces the prefilter's super-batch cut (UC_DRUN_MAX) or the per-position path.  Run on the GPU box after
kernel changes; the result line goes into DESIGN.md 2."""
import os, sys, time, traceback
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import torch  # noqa: F401
from oracle import oracle_py as O
import test_gpu_parity as T
first, n = int(sys.argv[1]), int(sys.argv[2])
fn = T.test_random_option_sets_against_the_oracle
fn = getattr(fn, "__wrapped__", fn)
bad, t0 = [], time.time()
for seed in range(first, first + n):
    env = {}
    if seed % 3 == 1: env = {"UC_DRUN_MAX": str(500 + 37 * (seed % 50))}
    if seed % 3 == 2 and seed % 2 == 0: env = {"UC_SIM_PER_POSITION": "1"}
    os.environ.update(env)
    try:
        fn(O, seed)
    except Exception as e:   # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", "".join(traceback.format_exception_only(type(e), e)).strip()[:400], flush=True)
    finally:
        for k in env: os.environ.pop(k, None)
print("property campaign: seeds %d..%d, %d failures %s, %.0f s" % (first, first + n - 1, len(bad), bad, time.time() - t0))
