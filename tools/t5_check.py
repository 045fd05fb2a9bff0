#!/usr/bin/env python3
"""tools/t5_check.py — HIP ProstT5 encoder vs the fp32 PyTorch restatement on seeded synthetic weights (run on the GPU box).
usage: t5_check.py [d_model n_heads d_ff n_layers]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)
import unicore_amd as U
from oracle import prostt5_ref as R

a = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else [128, 2, 512, 2]
cfg = R.default_config(d_model=a[0], n_heads=a[1], d_kv=128, d_ff=a[2], n_layers=a[3])
path = "/tmp/t5_%d_%d_%d_%d.gguf" % tuple(a)
R.write_synthetic_gguf(path, cfg, seed=0x5EED0005)
kv, w = R.read_gguf(path)
rng = np.random.default_rng(3)
seqs = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), L)) for L in (1, 5, 30, 62, 63, 64, 130, 300)] + ["MKTAYIAKQRXBZQISFVKSHFSRQ"]
enc = U.T5Encoder(path)
t0 = time.time(); codes, logits = enc.encode(seqs, logits=True); t1 = time.time()
worst = 0.0; agree = tot = 0
for s, c, lg in zip(seqs, codes, logits):
    rl, rc = R.forward(w, cfg, s, run_layers=None, part=3)
    err = np.abs(lg - rl).max() / max(np.abs(rl).max(), 1e-6)
    worst = max(worst, err)
    agree += int((c == rc).sum()); tot += len(rc)
    print("L=%4d  max|dlogit|/max|logit| = %.4f  argmax agree %d/%d" % (len(s), err, (c == rc).sum(), len(rc)))
print("worst rel err %.4f, argmax agreement %.4f, encode %.3fs, stats %s" % (worst, agree / tot, t1 - t0, enc.stats()))
