#!/usr/bin/env python3
"""3Di state statistics of the full-size SYNTHETIC ProstT5 model (fp32 torch restatement, CPU): state entropy, how often a state
repeats its predecessor, 6-mer diversity, 3Di identity of a homolog pair.  --calibrate prints the mean logit per state over 8
synthetic proteins — the vector oracle/prostt5_ref.py:FULL_MODEL_HEAD_CALIBRATION was taken from (run it on the UNcalibrated
model: pass --uncalibrated to write that model to /tmp first).  Needs ~12 GB RAM; a few minutes on 8 cores.
   python tools/t5_state_hist.py [--calibrate] [--uncalibrated]"""
import collections
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import prostt5_ref as R  # noqa: E402

LET = "ACDEFGHIKLMNPQRSTVWY"


def stats(strs, tag):
    allc = "".join(strs)
    p = np.array(sorted(collections.Counter(allc).values(), reverse=True)) / len(allc)
    runs = np.mean([np.mean([a == b for a, b in zip(s[:-1], s[1:])]) for s in strs])
    k6 = collections.Counter(s[i:i + 6] for s in strs for i in range(len(s) - 5))
    print("%-22s entropy %.2f bits  top3 %s  same-as-previous %.2f  distinct 6-mers %d of %d" % (tag, -(p * np.log2(p)).sum(), np.round(p[:3], 2), runs, len(k6), sum(k6.values())))


def main():
    cfg = R.default_config()
    path = "/tmp/t5_hist_%s.gguf" % ("raw" if "--uncalibrated" in sys.argv else "cal")
    if not os.path.exists(path):
        R.write_synthetic_gguf(path, cfg, seed=0x5EED0005 if "--uncalibrated" not in sys.argv else 0x5EED0005)
    _, w = R.read_gguf(path)
    w = dict(w)
    if "--uncalibrated" in sys.argv:
        w["cnn.conv2.bias"] = w["cnn.conv2.bias"] + np.asarray(R.FULL_MODEL_HEAD_CALIBRATION, np.float32)
    W = R.prepare(w)
    db = "/tmp/t5_hist_db"
    if not os.path.exists(db):
        subprocess.check_call([os.path.join(ROOT, "bin", "gen_synth"), db, "1", "0x5EED0005", "6000", "1.0"], stderr=subprocess.DEVNULL)
    aa = [e.decode() for e in open(db, "rb").read().split(b"\n\0")[:-1]]
    ss = [e.decode() for e in open(db + "_ss", "rb").read().split(b"\n\0")[:-1]]
    sel = [i for i in range(len(aa)) if 150 <= len(aa[i]) <= 400][:24]
    stats([ss[i] for i in sel[8:]], "gen_synth 3Di track")
    lgs = [R.forward(W, cfg, aa[i])[0] for i in sel]
    stats(["".join(LET[x] for x in l.argmax(1)) for l in lgs[8:]], "synthetic ProstT5")
    if "--calibrate" in sys.argv:
        print("mean logit per state over 8 proteins:", [round(float(x), 4) for x in np.concatenate(lgs[:8]).mean(0)])
    rng = np.random.default_rng(1)
    s = aa[sel[10]]
    m = list(s)
    for p in rng.choice(len(s), len(s) // 8, replace=False):
        m[p] = rng.choice(list(LET))
    a, b = lgs[10].argmax(1), R.forward(W, cfg, "".join(m))[0].argmax(1)
    print("3Di identity of a homolog with 12 %% substituted residues: %.2f; of an unrelated protein: %.2f" % ((a == b).mean(), (a[:150] == lgs[11].argmax(1)[:150]).mean()))


if __name__ == "__main__":
    main()
