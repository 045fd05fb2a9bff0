#!/usr/bin/env python3
"""tools/t5_gemm_ab.py — the GEMM kernels of the encoder on the same sequences (one process each; UC_T5_GEMM256 = 2: 256 x 256 tile in two phases per
K-step, persistent workgroups (the default); 1: 256 x 256 tile, one barrier per K-step; 0: 128 x 128 tile): predicted states and logits must be
IDENTICAL (same K order per output element), plus throughput.
usage: t5_gemm_ab.py [n_layers=4] [n_seqs=400]"""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

def work(out, nl, ns):
    import torch  # noqa: F401
    import unicore_amd as U
    from oracle import prostt5_ref as R
    cfg = R.default_config(n_layers=nl)
    path = "/tmp/prostt5_synth_%d.gguf" % nl
    if not os.path.exists(path):
        R.write_synthetic_gguf(path, cfg, seed=0x5EED0005)
    rng = np.random.default_rng(11)
    lens = np.clip(np.round(rng.lognormal(5.45, 0.76, ns)), 50, 2000).astype(int)
    seqs = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), int(L))) for L in lens]
    enc = U.T5Encoder(path)
    enc.encode(seqs[:32])
    s0 = enc.stats()
    codes, logits = enc.encode(seqs, logits=True)
    s1 = enc.stats()
    fl, ms = s1["flops"] - s0["flops"], s1["gpu_ms"] - s0["gpu_ms"]
    np.savez(out, codes=np.concatenate(codes), logits=np.concatenate([l.ravel() for l in logits]), tflops=fl / (ms * 1e-3) / 1e12, gpu_ms=ms)

if len(sys.argv) > 3:
    work(sys.argv[1], int(sys.argv[2]), int(sys.argv[3])); sys.exit(0)
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 400
res = {}
for name, env in (("twophase_256", {"UC_T5_GEMM256": "2"}), ("8wave_256", {"UC_T5_GEMM256": "1"}), ("128", {"UC_T5_GEMM256": "0"})):
    o = "/tmp/t5_ab_%s.npz" % name
    subprocess.check_call([sys.executable, os.path.abspath(__file__), o, str(nl), str(ns)], env=dict(os.environ, **env))
    res[name] = np.load(o)
b = res["128"]
out = {"codes_equal": bool(all((r["codes"] == b["codes"]).all() for r in res.values())),
       "logits_max_abs_diff": float(max(np.abs(r["logits"] - b["logits"]).max() for r in res.values())), "n_layers": nl, "n_seqs": ns}
for k, r in res.items():
    out["tflops_" + k] = float(r["tflops"]); out["gpu_ms_" + k] = float(r["gpu_ms"])
print(json.dumps(out))
