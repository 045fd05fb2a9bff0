#!/usr/bin/env python3
"""Regenerate tests/golden/t5_full_depth.npz: logits and 3Di codes of the fp32 PyTorch restatement (oracle/prostt5_ref.py) for
the FULL ProtT5-XL geometry — 24 blocks, d_model 1024, 32 x 128 heads, d_ff 16384, ProstT5 CNN head — with the seeded synthetic
weights `bench.py --config c5` uses (seed 0x5EED0005; the 2.4 GB GGUF is regenerated from the seed wherever it is needed: the
writer is deterministic).  The fixture lets the GPU box check the 24-block HIP encoder without running torch at full depth.
Run from the repo root (needs ~12 GB of RAM, a few minutes):  python tests/golden/make_t5_full_depth.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import prostt5_ref as R  # noqa: E402

SEED = 0x5EED0005
SEQS = ["M", "MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQAPILSRVGDGTQDNLSGAEKAVQVKVKALPDAQFEVVHSLAKWKRQTLGQHDFSAGEGLYTHMKALRPDEDRLSPLHSVYVDQWDWERVM",
        "ACDEFGHIKLMNPQRSTVWYXBZOUacdefghik"]
GGUF = os.environ.get("UC_T5_FULL_GGUF", "/tmp/uc_bench/prostt5_synth_24.gguf")


def ensure_gguf(path=GGUF):
    """the synthetic full-size model file (shared with bench.py --config c5 and the GPU tests)"""
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        R.write_synthetic_gguf(path + ".tmp%d" % os.getpid(), R.default_config(), seed=SEED)
        os.replace(path + ".tmp%d" % os.getpid(), path)
    return path


def main():
    cfg = R.default_config()
    _, w = R.read_gguf(ensure_gguf())
    out = {}
    for i, s in enumerate(SEQS):
        lg, codes = R.forward(w, cfg, s)                                          # the default head convention
        out["logits%d" % i] = lg.astype(np.float32)
        out["codes%d" % i] = codes
        lg, codes = R.forward(w, cfg, s, eos_in_head=False, uzob_to_x=True)       # the predict_3Di reading (UC_T5_EOS_IN_HEAD=0 UC_T5_KEEP_UZOB=0)
        out["logits%d_p3d" % i] = lg.astype(np.float32)
        out["codes%d_p3d" % i] = codes
    np.savez_compressed(os.path.join(HERE, "t5_full_depth.npz"), seqs=np.array(SEQS), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
