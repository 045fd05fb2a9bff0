#!/usr/bin/env python3
"""Regenerate tests/golden/t5_tiny.npz: logits and 3Di codes of the fp32 PyTorch restatement (oracle/prostt5_ref.py) for a
tiny ProstT5-shaped model with seeded synthetic weights (the GGUF itself is regenerated from the seed by the tests: the
writer is deterministic).  Run from the repo root:  python tests/golden/make_t5_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import prostt5_ref as R  # noqa: E402

CFG = dict(d_model=128, n_heads=2, d_kv=128, d_ff=512, n_layers=2)
SEED = 0x5EED0005
SEQS = ["M", "MKTAY", "MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQAPILSRVGDGTQDNLSGAEKAVQVKVKALPDAQFEVVHSLAKWKRQTLGQHDFSAGEGLYTHMKALRPDEDRLSPLHSVYVDQWDWERVMGDGERQFSTLKSTVEAIWAGIKATEAAVSEEFGLAPFLPDQIHFVHSQELLSRYPDLDAKGRERAIAKDLGAVFLVGIGGKLSDGHRHDVRAPDYDDWSTPSELGHAGLNGDILVWNPVLEDAFELSSMGIRVDADTLKHQLALTGDEDRLELEWHQALLRGEMPQTIGGGIGQSRLTMLLLQLPHIGQVQAGVWPAAVRESVPSLL",
        "ACDEFGHIKLMNPQRSTVWYXBZOU", "G" * 70]


def main():
    cfg = R.default_config(**CFG)
    path = "/tmp/t5_golden.gguf"
    R.write_synthetic_gguf(path, cfg, seed=SEED)
    _, w = R.read_gguf(path)
    out = {}
    for i, s in enumerate(SEQS):
        lg, codes = R.forward(w, cfg, s)                                          # the default head convention
        out["logits%d" % i] = lg.astype(np.float32)
        out["codes%d" % i] = codes
        lg, codes = R.forward(w, cfg, s, eos_in_head=False, uzob_to_x=True)       # the predict_3Di reading (UC_T5_EOS_IN_HEAD=0 UC_T5_KEEP_UZOB=0)
        out["logits%d_p3d" % i] = lg.astype(np.float32)
        out["codes%d_p3d" % i] = codes
    np.savez_compressed(os.path.join(HERE, "t5_tiny.npz"), seqs=np.array(SEQS), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
