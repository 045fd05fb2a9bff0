#!/usr/bin/env python3
"""tools/t5_bench.py — throughput of the ProstT5 encoder at ProtT5-XL geometry (24 blocks, 1024 / 32 x 128 / 16384) with
seeded random-init weights (no real weights can be shipped) on protein-length-distributed synthetic sequences.
usage: t5_bench.py [n_layers=24] [n_seqs=2000] [batch_tokens=65536]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import unicore_amd as U
from oracle import prostt5_ref as R

nl = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
if len(sys.argv) > 3:
    os.environ["UC_T5_BATCH_TOKENS"] = sys.argv[3]
cfg = R.default_config(n_layers=nl)
path = "/tmp/prostt5_synth_%d.gguf" % nl
if not os.path.exists(path):
    t0 = time.time(); R.write_synthetic_gguf(path, cfg, seed=0x5EED0005); print("wrote %s in %.1fs (%.2f GB)" % (path, time.time() - t0, os.path.getsize(path) / 1e9), file=sys.stderr)
rng = np.random.default_rng(11)
lens = np.clip(np.round(rng.lognormal(5.45, 0.76, ns)), 50, 2000).astype(int)          # the length model of tools/gen_synth.c
seqs = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), int(L))) for L in lens]
t0 = time.time(); enc = U.T5Encoder(path); t_load = time.time() - t0
enc.encode(seqs[:64])                                                                  # warm-up (allocations)
s0 = enc.stats()
t0 = time.time(); enc.encode(seqs); wall = time.time() - t0
s1 = enc.stats()
fl, ms, tok = s1["flops"] - s0["flops"], s1["gpu_ms"] - s0["gpu_ms"], s1["n_tokens"] - s0["n_tokens"]
print(json.dumps({"n_layers": nl, "n_seqs": ns, "residues": int(lens.sum()), "tokens": tok, "load_s": t_load, "wall_s": wall, "gpu_ms": ms,
                  "tflops": fl / (ms * 1e-3) / 1e12, "mfma_frac_of_2500": fl / (ms * 1e-3) / 2.5e15, "residues_per_s": int(lens.sum()) / wall}))
