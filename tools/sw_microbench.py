#!/usr/bin/env python3
"""Gapped-kernel microbenchmark through the C ABI: uniform lengths, many pairs per query — isolates the per-cell
cost of the int32 and the packed 16-bit kernel from workload effects (ragged lengths, few pairs per query)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unicore_amd as U

def run(L, nq, per_q, kernel, mode=0, seed=1):
    rng = np.random.default_rng(seed)
    nt = 4096
    n = nq + nt
    s3 = rng.integers(0, 20, (n, L), dtype=np.uint8)
    sa = rng.integers(0, 20, (n, L), dtype=np.uint8)
    # make targets related to queries so scores are realistic
    for t in range(nt):
        q = t % nq
        keep = rng.random(L) > 0.3
        s3[nq + t][keep] = s3[q][keep]; sa[nq + t][keep] = sa[q][keep]
    off = np.arange(n + 1, dtype=np.uint64) * L
    e = U.Engine("-c 0.8 --sw-kernel " + kernel, verbosity=1)
    e.set_db(off, s3.reshape(-1), sa.reshape(-1))
    q = np.repeat(np.arange(nq, dtype=np.uint32), per_q)
    t = (nq + (np.arange(nq * per_q) * 7919) % nt).astype(np.uint32)
    e.sw(mode, q[:64], t[:64])          # warm-up
    e.reset_stats()
    s, qe, te = e.sw(mode, q, t)
    st = e.stats()
    cells = len(q) * L * L
    return st["sw_kernel_ms"], cells / st["sw_kernel_ms"] / 1e6, st["n_pk_reruns"], int(s.sum())

if __name__ == "__main__" and len(sys.argv) > 1:      # single configuration (for rocprofv3): L kernel mode
    L, kern, mode = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    print(L, kern, mode, run(L, 256, max(16, 200000 // L), kern, mode))
elif __name__ == "__main__":
    for L in (120, 250, 500, 1000):
        for mode in (0, 1):
            r = {k: run(L, 256, max(16, 200000 // L), k, mode) for k in ("i32", "pk16")}
            print("L=%4d mode=%d  i32: %8.2f ms %7.1f GCUPS | pk16: %8.2f ms %7.1f GCUPS reruns=%d | speedup %.2f  same=%s"
                  % (L, mode, r["i32"][0], r["i32"][1], r["pk16"][0], r["pk16"][1], r["pk16"][2], r["i32"][0] / r["pk16"][0], r["i32"][3] == r["pk16"][3]))
