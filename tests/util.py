"""Shared helpers for the tests: seeded synthetic DBs, oracle<->engine option mirroring."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LET = "ACDEFGHIKLMNPQRSTVWY"


def ensure_tools():
    gen = os.path.join(ROOT, "bin", "gen_synth")
    if not os.path.exists(gen):
        subprocess.check_call(["make", "-C", ROOT, "tools"], stdout=subprocess.DEVNULL)
    return gen


DB_SUFFIXES = ("", "_ss", "_h", ".index", "_ss.index", "_h.index", ".dbtype", "_ss.dbtype", "_h.dbtype", ".lookup", ".map")
_DB_CACHE = {}          # (proteomes, seed, families, scale) -> prefix of the copy generated first in this session


def gen_synth_db(prefix, n_proteomes, seed, n_families, len_scale):
    """the seeded synthetic proteome database at `prefix`.  At-size databases (>= 50 proteomes: 4-70 s of generation, up to 4.5 GB) are generated ONCE per test
    session: test_configs_gpu.py and test_workflow_gpu.py ask for the same (proteomes, seed) several times, and every later request gets hard links to the
    first copy's files (the files are never modified in place; a link to a removed directory's files stays valid)."""
    key = (int(n_proteomes), int(seed), int(n_families), float(len_scale))
    src = _DB_CACHE.get(key)
    if src and all(os.path.exists(src + s) for s in DB_SUFFIXES):
        try:
            for s in DB_SUFFIXES:
                if os.path.exists(prefix + s):
                    os.remove(prefix + s)
                os.link(src + s, prefix + s)
            return prefix
        except OSError:
            pass                      # another file system: generate
    subprocess.check_call([ensure_tools(), prefix, str(n_proteomes), hex(seed), str(n_families), str(len_scale)],
                          stderr=subprocess.DEVNULL)
    if n_proteomes >= 50:
        _DB_CACHE[key] = prefix
    return prefix


def family_db(seed, n_fam=12, members=6, lmin=20, lmax=260, extra=(), sub3=0.15, suba=0.3, indel=0.02, with_x=True):
    """Small family-structured DB as code arrays (s3 list, sa list).  `extra` = extra sequence lengths of
    unrelated random sequences (used to reach long-sequence kernel classes)."""
    rng = np.random.default_rng(seed)
    s3, sa = [], []
    for f in range(n_fam):
        L = int(rng.integers(lmin, lmax + 1))
        a3, aa = rng.integers(0, 20, L, dtype=np.uint8), rng.integers(0, 20, L, dtype=np.uint8)
        for m in range(members):
            keep = rng.random(L) >= indel
            m3, ma = a3[keep].copy(), aa[keep].copy()
            mut3, muta = rng.random(len(m3)) < sub3, rng.random(len(ma)) < suba
            m3[mut3] = rng.integers(0, 20, int(mut3.sum()), dtype=np.uint8)
            ma[muta] = rng.integers(0, 20, int(muta.sum()), dtype=np.uint8)
            if m == members - 1 and len(m3) > 30:      # one truncated member per family
                cut = int(len(m3) * 0.6)
                m3, ma = m3[:cut], ma[:cut]
            if with_x and m == 1 and len(m3) > 12:      # an X in both tracks
                m3[7] = 20
                ma[3] = 20
            s3.append(m3)
            sa.append(ma)
    for L in extra:
        s3.append(rng.integers(0, 20, L, dtype=np.uint8))
        sa.append(rng.integers(0, 20, L, dtype=np.uint8))
    # edge cases: shorter than the k-mer span, length 1, all-X
    for L in (1, 5, 9):
        s3.append(rng.integers(0, 20, L, dtype=np.uint8))
        sa.append(rng.integers(0, 20, L, dtype=np.uint8))
    s3.append(np.full(15, 20, np.uint8))
    sa.append(np.full(15, 20, np.uint8))
    return s3, sa


def flat(s3, sa):
    off = np.zeros(len(s3) + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in s3])
    return off, np.concatenate(s3), np.concatenate(sa)


def write_db(prefix, s3, sa, names=None):
    """Write code arrays as an MMseqs-style DB (same files gen_synth produces)."""
    n = len(s3)
    names = names or ["unicore_%010x" % (i * 7919 + 1) for i in range(n)]
    letters = np.frombuffer((LET + "X").encode(), np.uint8)
    for suffix, seqs in (("", sa), ("_ss", s3)):
        offv = 0
        with open(prefix + suffix, "wb") as f, open(prefix + suffix + ".index", "w") as ix:
            for i, s in enumerate(seqs):
                f.write(letters[np.asarray(s)].tobytes() + b"\n\0")
                ix.write("%d\t%d\t%d\n" % (i, offv, len(s) + 2))
                offv += len(s) + 2
        with open(prefix + suffix + ".dbtype", "wb") as f:
            f.write((0).to_bytes(4, "little"))
    offv = 0
    with open(prefix + "_h", "wb") as f, open(prefix + "_h.index", "w") as ix, open(prefix + ".lookup", "w") as lk:
        for i, nm in enumerate(names):
            f.write(nm.encode() + b"\n\0")
            ix.write("%d\t%d\t%d\n" % (i, offv, len(nm) + 2))
            lk.write("%d\t%s\t0\n" % (i, nm))
            offv += len(nm) + 2
    with open(prefix + "_h.dbtype", "wb") as f:
        f.write((12).to_bytes(4, "little"))
    return names


def oracle_params(O, opts=""):
    """Mirror a Foldseek-style option string onto oracle params (test-side restatement of the option
    semantics: -c, --cov-mode, -e, -s/--k-score, --max-seqs, --min-ungapped-score ...)."""
    kw = {}
    tok = opts.split()
    i = 0
    sens = 4.0
    kscore = None
    while i < len(tok):
        f, v = tok[i], tok[i + 1] if i + 1 < len(tok) else None
        if f == "-c": kw["cov"] = float(v)
        elif f == "--cov-mode": kw["cov_mode"] = int(v)
        elif f == "-e": kw["evalue"] = float(v)
        elif f == "-s": sens = float(v)
        elif f == "--k-score": kscore = int(v)
        elif f == "--max-seqs": kw["max_seqs"] = int(v)
        elif f == "--min-ungapped-score": kw["min_ungapped"] = int(v)
        elif f == "--min-seq-id": kw["min_seq_id"] = float(v)
        elif f == "--min-diag-hits": kw["min_diag_hits"] = int(v)
        elif f == "--rev-correction": kw["rev_correction"] = int(v)
        elif f == "--gap-open": kw["gap_open"] = int(v)
        elif f == "--gap-extend": kw["gap_ext"] = int(v)
        elif f == "--mat-bit-factor-3di": kw["bit_factor_3di"] = float(v)
        elif f == "--mat-bit-factor-aa": kw["bit_factor_aa"] = float(v)
        elif f == "--comp-bias-corr": kw["_cb"] = int(v)
        elif f == "--comp-bias-corr-scale": kw["_cbs"] = float(v)
        elif f == "--length-gate": kw["len_gate"] = int(v)     # rule UC-1/L
        elif f in ("--sw-kernel", "--sym-dedup"): pass      # engine-side execution choices, no effect on results
        else: raise ValueError(f)
        i += 2
    cb, cbs = kw.pop("_cb", 0), kw.pop("_cbs", 1.0)
    if cb:
        kw["comp_bias_milli"] = int(round(cbs * 1000))       # rule UC-1/B: scale in thousandths
    p = O.default_params(**kw)
    if kscore is None:
        diag = sum(p.S3[a * 21 + a] for a in range(20))
        kscore = int(np.floor(6 * diag / 20.0 + 3.0 - 2.0 * sens + 0.5))
    p.kmer_thr = kscore
    return p


def tsv_invariants(tsv_path, names):
    """The requirements the reference's consumer puts on clust.tsv (src/modules/profile.rs:50-55,79-84;
    SURVEY.md 8a-R10): two columns, contiguous representatives, every DB name exactly once in col 1,
    every representative has its own rep\trep row first, col0 subset of col1."""
    rows = [l.rstrip("\n").split("\t") for l in open(tsv_path)]
    assert all(len(r) == 2 for r in rows)
    members = [r[1] for r in rows]
    assert sorted(members) == sorted(names), "every sequence must appear exactly once as a member"
    seen, prev = set(), None
    for rep, mem in rows:
        if rep != prev:
            assert rep not in seen, "rows of one representative must be contiguous"
            seen.add(rep)
            assert mem == rep, "representative row must come first"
            prev = rep
    assert seen <= set(members)
    return rows
