#!/usr/bin/env python3
"""Regenerate tests/golden/c1/ — BASELINE configs[0] as a parity-test case: the first 5 proteomes of the reference's
example/data (AA FASTA, read from /root/reference in THIS container; the derived DB files are the committed fixture),
written in the format `unicore createdb` leaves on disk (names unicore_<md5(aa)[:10]>, <db>.map lines
"name\\tspecies\\toriginal header", src/modules/createdb.rs:86-108).  ProstT5 weights are not obtainable offline, so the
3Di track is a DOCUMENTED STAND-IN derived deterministically from the AA track - the fixture pins
the plumbing and the engine on real protein lengths / compositions, not real structures.  Expected output: the oracle's
clust.tsv for `-c 0.8 --single-step-clustering` and clust_workflow.tsv for a bare `-c 0.8` (pre-step + 3-step cascade).   Run from the repo root:  python tests/golden/make_c1.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

SRC = "/root/reference/example/data"
LET = "ACDEFGHIKLMNPQRSTVWY"


def read_fasta(path):
    name, seq, out = None, [], []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                out.append((name, "".join(seq)))
            name, seq = line[1:], []
        else:
            seq.append(line.strip())
    if name is not None:
        out.append((name, "".join(seq)))
    return out


def main():
    files = sorted(f for f in os.listdir(SRC) if f.endswith(".fa"))[:5]
    seqs, maplines = {}, []
    for f in files:
        species = os.path.splitext(f)[0]
        for key, value in read_fasta(os.path.join(SRC, f)):
            if len(value) < 2:
                continue
            key = "".join("_" if c.isspace() else c for c in key)
            name = "unicore_" + hashlib.md5(value.encode()).hexdigest()[:10]
            seqs[name] = value
            maplines.append("%s\t%s\t%s\n" % (name, species, key))
    names = sorted(seqs)
    lut = np.full(256, 20, np.uint8)
    for i, c in enumerate(LET):
        lut[ord(c)] = i
    sa, s3 = [], []
    for n in names:
        a = lut[np.frombuffer(seqs[n].encode(), np.uint8)]
        prev, nxt = np.roll(a, 1), np.roll(a, -1)
        t = ((a.astype(np.int64) * 7 + nxt * 3 + prev) % 20).astype(np.uint8)      # stand-in 3Di track
        t[a == 20] = 20
        sa.append(a)
        s3.append(t)
    db = os.path.join(HERE, "c1", "db")
    util.write_db(db, s3, sa, names)
    open(db + ".map", "w").writelines(maplines)
    odb = O.OracleDb(db)
    p = util.oracle_params(O, "-c 0.8")
    r = O.cluster(odb, p, threads=4, dumps=False)
    O.write_tsv(os.path.join(HERE, "c1", "clust.tsv"), odb, r["assign"])
    # what a bare "-c 0.8" runs by default: linear-time pre-step (20 k-mers per sequence) + 3-step cascade
    rw = O.cluster_workflow(odb, p, O.cascade_thresholds(p, 4.0, 3), linclust_m=20, threads=4)
    O.write_tsv(os.path.join(HERE, "c1", "clust_workflow.tsv"), odb, rw["assign"])
    print("default workflow:", rw["round_sizes"].tolist(), "sequences per round,", rw["counts"]["n_clusters"], "clusters")
    print(len(files), "proteomes,", odb.n, "sequences,", r["counts"]["n_alignments"], "alignments,", r["counts"]["n_clusters"], "clusters")


if __name__ == "__main__":
    main()
