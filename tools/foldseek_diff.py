#!/usr/bin/env python3
"""tools/foldseek_diff.py — the day a real `foldseek` binary is available: where does UC-1 (this repository's frozen spec, restated by
oracle/uc_oracle.c and executed by the HIP engine) first part ways with it?

    python tools/foldseek_diff.py <seqDB> [--foldseek PATH] [--options "-c 0.8"] [--keep DIR]

What it does (reference call site: /root/reference/src/modules/cluster.rs:45-56):
  1. runs `foldseek cluster <seqDB> <out> <tmp> <options> --single-step-clustering -v 3 --remove-tmp-files 0` (and, with --workflow,
     the default cascaded workflow too), so that the intermediate MMseqs-style databases stay in <tmp>;
  2. finds them in the tmp tree — prefilter results (`pref*`), alignment results (`aln*`), cluster results (`clu*`) — and parses
     them (entries "...\\n\\0", index "key\\toffset\\tlength"; prefilter lines `target\\tscore\\tdiagonal`, alignment lines
     `target\\tbits\\tseqId\\tevalue\\tqStart\\tqEnd\\tqLen\\ttStart\\ttEnd\\ttLen...`, cluster lines = member keys; SURVEY.md App. B,
     layout EXT-UNVERIFIED: the globs are deliberately loose and every database found is reported);
  3. runs the oracle's single step on the same database with the same options and compares STAGE BY STAGE, stopping at the first
     one that differs: (E2-E4) the per-query hit lists — target sets, then ungapped score and diagonal of the common targets;
     (E5/E6) the accepted pairs and, for common pairs, start / end coordinates (bit scores and E-values are printed side by side:
     UC-1 uses a Karlin-Altschul E-value, Foldseek a fitted one — INTEGRATION.md §D); (E7) the representative of every sequence;
  4. prints, per stage, the number of queries that agree and the first few that do not, with both sides' records.

--self-test needs no Foldseek: it writes the oracle's own dumps as MMseqs-style databases into a fake tmp tree, parses them back and
diffs (zero differences expected), then plants one difference per stage and checks that it is found (tests/test_oracle_kat.py runs it).
"""
import argparse
import glob
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")


# ---------------------------------------------------------------------------------------------- MMseqs-style result databases
def read_result_db(prefix):
    """{query key: [fields of every line]} of one result database; split databases (prefix.0, prefix.1, ...) are concatenated"""
    idx_files = [prefix + ".index"] if os.path.exists(prefix + ".index") else []
    out = {}
    for ix in idx_files:
        data_files = [prefix] if os.path.exists(prefix) else sorted(glob.glob(prefix + ".[0-9]*"), key=lambda f: int(f.rsplit(".", 1)[1]))
        blob = b"".join(open(f, "rb").read() for f in data_files)
        for line in open(ix):
            k, off, ln = line.split()[:3]
            body = blob[int(off):int(off) + int(ln)].rstrip(b"\0")
            out[int(k)] = [l.split("\t") for l in body.decode(errors="replace").split("\n") if l]
    return out


def write_result_db(prefix, entries):
    """entries: {key: [line, ...]} (the --self-test writes what it later parses)"""
    off = 0
    with open(prefix, "wb") as f, open(prefix + ".index", "w") as ix:
        for k in sorted(entries):
            body = ("".join(l + "\n" for l in entries[k])).encode() + b"\0"
            f.write(body)
            ix.write("%d\t%d\t%d\n" % (k, off, len(body)))
            off += len(body)


def find_dbs(tmp):
    """result databases of one run by kind, shallowest first (the single-step run has one of each; the workflow one per step)"""
    found = {"pref": [], "aln": [], "clu": []}
    for dirpath, _, files in os.walk(tmp):
        for f in files:
            if not f.endswith(".index"):
                continue
            base = f[:-6]
            for kind in found:
                if base.startswith(kind) and not base.endswith(("_h", "_tmp")):
                    found[kind].append(os.path.join(dirpath, base))
    for kind in found:
        found[kind].sort(key=lambda p: (p.count(os.sep), p))
    return found


# ---------------------------------------------------------------------------------------------- the oracle's side
def oracle_single_step(db, options):
    import util
    from oracle import oracle_py as O
    odb = O.OracleDb(db)
    p = util.oracle_params(O, options)
    res = O.cluster(odb, p, threads=0, dumps=True)
    idx = np.loadtxt(db + ".index", dtype=np.int64, ndmin=2)
    keys = idx[np.argsort(idx[:, 0], kind="stable")][:, 0]
    lens = np.diff(odb.offsets()).astype(np.int64)
    total = int(lens.sum())
    pref, aln = {}, {}
    for q in range(odb.n):
        c = int(res["hit_cnt"][q])
        h, a = res["hits"][q, :c], res["aln"][q, :c]
        pref[int(keys[q])] = {int(keys[t]): (int(s), int(d)) for t, s, d in zip(h["t"], h["score"], h["diag"])}
        acc = {}
        for k in range(c):
            if a[k]["accepted"]:
                bits = int((p.lambda_ * float(a[k]["corrected"]) - np.log(p.K)) / np.log(2.0))
                ev = p.K * float(lens[q]) * float(total) * np.exp(-p.lambda_ * float(a[k]["corrected"]))
                acc[int(keys[int(h[k]["t"])])] = dict(bits=bits, evalue=ev, qstart=int(a[k]["qstart"]), qend=int(a[k]["qend"]),
                                                      tstart=int(a[k]["tstart"]), tend=int(a[k]["tend"]), raw=int(a[k]["score"]))
        aln[int(keys[q])] = acc
    clu = {}
    for i, r in enumerate(res["assign"]):
        clu.setdefault(int(keys[int(r)]), []).append(int(keys[i]))
    return pref, aln, clu


# ---------------------------------------------------------------------------------------------- the comparison
def parse_pref(db):
    return {q: {int(f[0]): (int(f[1]), int(f[2])) for f in lines if len(f) >= 3} for q, lines in db.items()}


def parse_aln(db):
    out = {}
    for q, lines in db.items():
        d = {}
        for f in lines:
            if len(f) >= 10:   # Foldseek / MMseqs2 coordinates are 1-based in the alignment DB when written by convertalis, 0-based here: both are tried below
                d[int(f[0])] = dict(bits=int(float(f[1])), evalue=float(f[3]), qstart=int(f[4]), qend=int(f[5]), tstart=int(f[7]), tend=int(f[8]))
        out[q] = d
    return out


def parse_clu(db):
    return {rep: [int(f[0]) for f in lines] for rep, lines in db.items()}


def diff_stage(name, ours, theirs, cmp_fields, show=5):
    """ours / theirs: {query: {target: record}}.  Returns the number of queries that differ."""
    bad = []
    for q in sorted(set(ours) | set(theirs)):
        a, b = ours.get(q, {}), theirs.get(q, {})
        if set(a) != set(b):
            bad.append((q, "target sets differ: only UC-1 %s, only Foldseek %s" % (sorted(set(a) - set(b))[:6], sorted(set(b) - set(a))[:6])))
            continue
        for t in a:
            ra, rb = a[t], b[t]
            da = [ra[f] for f in cmp_fields] if isinstance(ra, dict) else list(ra)
            db_ = [rb[f] for f in cmp_fields] if isinstance(rb, dict) else list(rb)
            if da != db_:
                bad.append((q, "target %d: UC-1 %s, Foldseek %s" % (t, ra, rb)))
                break
    n = len(set(ours) | set(theirs))
    print("%-28s %d of %d queries identical" % (name, n - len(bad), n))
    for q, why in bad[:show]:
        print("    query %d: %s" % (q, why))
    return len(bad)


def compare(ours, theirs):
    """(pref, aln, clu) triples -> index of the first diverging stage (0..2) or -1"""
    opref, oaln, oclu = ours
    tpref, taln, tclu = theirs
    if diff_stage("E2-E4 prefilter hit lists", opref, tpref, None):
        return 0
    if diff_stage("E5-E6 accepted alignments", oaln, taln, ("qstart", "qend", "tstart", "tend")):
        return 1
    orep = {m: r for r, ms in oclu.items() for m in ms}
    trep = {m: r for r, ms in tclu.items() for m in ms}
    bad = [m for m in sorted(set(orep) | set(trep)) if orep.get(m) != trep.get(m)]
    print("%-28s %d of %d sequences have the same representative" % ("E7 set cover", len(orep) - len(bad), len(orep)))
    for m in bad[:5]:
        print("    sequence %d: UC-1 representative %s, Foldseek %s" % (m, orep.get(m), trep.get(m)))
    return 2 if bad else -1


def to_result_dbs(ours, tmp):
    pref, aln, clu = ours
    os.makedirs(tmp, exist_ok=True)
    write_result_db(os.path.join(tmp, "pref"), {q: ["%d\t%d\t%d" % (t, s, d) for t, (s, d) in sorted(h.items())] for q, h in pref.items()})
    write_result_db(os.path.join(tmp, "aln"), {q: ["%d\t%d\t1.000\t%.3E\t%d\t%d\t0\t%d\t%d\t0" % (t, r["bits"], r["evalue"], r["qstart"], r["qend"], r["tstart"], r["tend"])
                                                   for t, r in sorted(a.items())] for q, a in aln.items()})
    write_result_db(os.path.join(tmp, "clu"), {r: ["%d" % m for m in ms] for r, ms in clu.items()})


def load_tmp(tmp):
    dbs = find_dbs(tmp)
    for kind, paths in dbs.items():
        print("%s databases found: %s" % (kind, [os.path.relpath(p, tmp) for p in paths] or "none"))
    if not (dbs["pref"] and dbs["aln"] and dbs["clu"]):
        raise SystemExit("the tmp tree does not hold a prefilter, an alignment and a cluster database (was --remove-tmp-files 0 honoured?)")
    return parse_pref(read_result_db(dbs["pref"][0])), parse_aln(read_result_db(dbs["aln"][0])), parse_clu(read_result_db(dbs["clu"][0]))


def self_test():
    import util
    work = tempfile.mkdtemp(prefix="fsdiff_")
    try:
        s3, sa = util.family_db(5, n_fam=6, members=4)
        db = os.path.join(work, "db")
        util.write_db(db, s3, sa)
        ours = oracle_single_step(db, "-c 0.8")
        to_result_dbs(ours, os.path.join(work, "tmp", "1234567"))
        assert compare(ours, load_tmp(os.path.join(work, "tmp"))) == -1
        # plant one difference per stage: the differ must name that stage
        for stage in (0, 1, 2):
            pref, aln, clu = [dict((k, dict(v) if isinstance(v, dict) else list(v)) for k, v in d.items()) for d in ours]
            if stage == 0:
                q = next(k for k, v in pref.items() if v)
                t = next(iter(pref[q]))
                pref[q][t] = (pref[q][t][0] + 1, pref[q][t][1])
            elif stage == 1:
                q = next(k for k, v in aln.items() if v)
                t = next(iter(aln[q]))
                aln[q][t] = dict(aln[q][t], qend=aln[q][t]["qend"] + 1)
            else:
                r = next(k for k, v in clu.items() if len(v) > 1)
                m = [x for x in clu[r] if x != r][0]
                clu[r] = [x for x in clu[r] if x != m]
                clu[m] = [m]
            shutil.rmtree(os.path.join(work, "tmp"))
            to_result_dbs((pref, aln, clu), os.path.join(work, "tmp", "1234567"))
            assert compare(ours, load_tmp(os.path.join(work, "tmp"))) == stage, stage
        print("self-test ok")
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("db", nargs="?")
    ap.add_argument("--foldseek", default=shutil.which("foldseek"))
    ap.add_argument("--options", default="-c 0.8")
    ap.add_argument("--keep")
    ap.add_argument("--self-test", action="store_true")
    a = ap.parse_args()
    if a.self_test:
        return self_test()
    if not a.db:
        ap.error("a sequence database is needed (or --self-test)")
    shim = os.path.realpath(os.path.join(ROOT, "bin", "foldseek"))
    if not a.foldseek or os.path.realpath(a.foldseek) == shim:
        raise SystemExit("no real foldseek binary found (PATH holds none, or only this repository's shim): nothing to diff against")
    work = a.keep or tempfile.mkdtemp(prefix="fsdiff_")
    os.makedirs(work, exist_ok=True)
    tmp = os.path.join(work, "tmp")
    v = subprocess.run([a.foldseek, "version"], capture_output=True, text=True).stdout.strip()
    print("foldseek %s (%s)" % (v, a.foldseek))
    # flags of this repository's optional rules steer the ORACLE side only: Foldseek does not know them
    own = {"--mat-bit-factor-3di", "--mat-bit-factor-aa", "--min-score-table", "--length-gate"}
    tok, theirs_opts, i = a.options.split(), [], 0
    while i < len(tok):
        if tok[i] in own: i += 2; continue
        theirs_opts.append(tok[i]); i += 1
    subprocess.check_call([a.foldseek, "cluster", a.db, os.path.join(work, "out_cluster"), tmp] + theirs_opts +
                          ["--single-step-clustering", "-v", "3", "--remove-tmp-files", "0"])
    theirs = load_tmp(tmp)
    ours = oracle_single_step(a.db, a.options)
    stage = compare(ours, theirs)
    print("first diverging stage: %s" % (["E2-E4 prefilter", "E5-E6 alignment", "E7 set cover"][stage] if stage >= 0 else "none — identical at every stage"))
    # the optional rules (all default off) that move each stage: try them one at a time through --options
    hints = {0: "stage 1 rules: --mat-bit-factor-3di 2.1 --mat-bit-factor-aa 1.4 (UC-1/M), --comp-bias-corr 1 (UC-1/B)",
             1: "stage 2 rules: --length-gate 1 (UC-1/L: pairs missing on Foldseek's side whose lengths rule the coverage out), --min-score-table FILE (UC-1/E)"}
    if stage in hints:
        print("switchable " + hints[stage])
    if not a.keep:
        shutil.rmtree(work, ignore_errors=True)
    return 0 if stage < 0 else 1


if __name__ == "__main__":
    sys.exit(main() or 0)
