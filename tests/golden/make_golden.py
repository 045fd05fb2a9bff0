#!/usr/bin/env python3
"""Regenerate tests/golden/ — ORACLE-GENERATED fixtures (the reference holds no golden vector for this path,
see oracle/uc_oracle.h "PARITY UNPINNED").  Inputs: a tiny seeded synthetic proteome set written by
tools/gen_synth.c in the exact DB format `unicore createdb` leaves on disk.  Expected outputs: the oracle's
per-stage dumps and clust.tsv for two option strings.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

CASES = {"default": "-c 0.8", "sensitive": "-c 0.5 -s 6 --max-seqs 8 -e 1e-4"}


def main():
    db = os.path.join(HERE, "db")
    subprocess.check_call(["make", "-C", ROOT, "tools", "oracle"], stdout=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(ROOT, "bin", "gen_synth"), db, "5", "0x601D", "32", "0.4"])
    odb = O.OracleDb(db)
    for name, opts in CASES.items():
        p = util.oracle_params(O, opts)
        r = O.cluster(odb, p, threads=4)
        cnt = r["hit_cnt"]
        hits = np.concatenate([r["hits"][q, : cnt[q]] for q in range(odb.n)])
        aln = np.concatenate([r["aln"][q, : cnt[q]] for q in range(odb.n)])
        np.savez_compressed(os.path.join(HERE, "expected_%s.npz" % name), options=np.array(opts), hit_cnt=cnt, hits=hits,
                            aln=aln, assign=r["assign"], counts=np.array([r["counts"][k] for k in sorted(r["counts"])], np.uint64),
                            count_names=np.array(sorted(r["counts"])))
        O.write_tsv(os.path.join(HERE, "clust_%s.tsv" % name), odb, r["assign"])
        print(name, opts, "->", odb.n, "seqs,", int(cnt.sum()), "alignments,", len(set(r["assign"].tolist())), "clusters")
    # cascade (E8) and search (8f rank 3) fixtures on the same DB: the golden DB searched against itself
    p = util.oracle_params(O, "-c 0.8")
    thr = O.cascade_thresholds(p, 4.0, 3)
    rc = O.cluster_cascade(odb, p, thr, threads=4)
    O.write_tsv(os.path.join(HERE, "clust_cascade3.tsv"), odb, rc["assign"])
    print("cascade3", thr, "->", rc["round_sizes"].tolist(), "sequences per round,", rc["counts"]["n_clusters"], "clusters")
    rw = O.cluster_workflow(odb, p, thr, linclust_m=20, threads=4)
    O.write_tsv(os.path.join(HERE, "clust_linclust_cascade3.tsv"), odb, rw["assign"])
    print("linclust + cascade3 ->", rw["round_sizes"].tolist(), "sequences per round,", rw["counts"]["n_clusters"], "clusters")
    ps = util.oracle_params(O, "-e 10 --max-seqs 1000 -c 0.8")
    rs = O.search(odb, odb, ps, threads=4)
    O.write_m8(os.path.join(HERE, "search_self.m8"), odb, odb, ps, rs)
    print("search_self ->", int(rs["counts"]["n_edges"]), "rows")


if __name__ == "__main__":
    main()
