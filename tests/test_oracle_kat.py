"""Known-answer and property tests that PIN THE ORACLE (CPU only).

The reference holds no golden vector for this path (parity unpinned, oracle/uc_oracle.h), so the oracle is
pinned by (a) hand-computable cases, (b) an independent pure-Python restatement of every stage on small
inputs, (c) the committed fixtures in tests/golden/ (test_golden.py).
"""
import ctypes as C
import itertools
import os

import numpy as np
import pytest

import util
from oracle import oracle_py as O

ROOT = util.ROOT

LET = "ACDEFGHIKLMNPQRSTVWY"


@pytest.fixture(scope="module")
def p():
    return O.default_params()


def S3(p, a, b):
    return p.S3[int(a) * 21 + int(b)]


def SA(p, a, b):
    return p.SA[int(a) * 21 + int(b)]


def enc(s):
    return O.encode(s)


# ------------------------------------------------------------------ alphabet / matrices
def test_letter_codes_and_matrix_values(p):
    for i, c in enumerate(LET):
        assert O.lib().uco_letter_code(c.encode()) == i and O.lib().uco_letter_code(c.lower().encode()) == i
    for c in "XBZJOU*-":
        assert O.lib().uco_letter_code(c.encode()) == 20
    # BLOSUM62 spot values (Henikoff & Henikoff 1992)
    idx = {c: i for i, c in enumerate(LET)}
    for a, b, v in [("A", "A", 4), ("W", "W", 11), ("C", "C", 9), ("W", "C", -2), ("D", "E", 2), ("I", "V", 3), ("G", "I", -4), ("H", "Y", 2)]:
        assert SA(p, idx[a], idx[b]) == v == SA(p, idx[b], idx[a])
    assert SA(p, 20, 0) == 0 and SA(p, 20, 20) == -1          # X column of BLOSUM62
    m3 = np.array(p.S3[:]).reshape(21, 21)
    assert (m3 == m3.T).all() and (np.diag(m3)[:20] >= 4).all() and (m3[20] == -1).all()


# ------------------------------------------------------------------ independent restatements (pure Python)
def py_sw(q3, qa, t3, ta, p):
    """textbook affine local alignment + the frozen tie-break (smallest tEnd, then smallest qEnd)"""
    NEG = -10 ** 9
    lq, lt = len(q3), len(t3)
    H = [[0] * (lt + 1) for _ in range(lq + 1)]
    E = [[NEG] * (lt + 1) for _ in range(lq + 1)]
    F = [[NEG] * (lt + 1) for _ in range(lq + 1)]
    for i in range(1, lq + 1):
        for j in range(1, lt + 1):
            s = S3(p, q3[i - 1], t3[j - 1]) + SA(p, qa[i - 1], ta[j - 1])
            E[i][j] = max(E[i][j - 1] - p.gap_ext, H[i][j - 1] - p.gap_open)
            F[i][j] = max(F[i - 1][j] - p.gap_ext, H[i - 1][j] - p.gap_open)
            H[i][j] = max(0, H[i - 1][j - 1] + s, E[i][j], F[i][j])
    best = max(max(r) for r in H)
    if best == 0:
        return 0, -1, -1
    for j in range(1, lt + 1):
        for i in range(1, lq + 1):
            if H[i][j] == best:
                return best, i - 1, j - 1


def py_ungapped(q3, t3, d, p):
    run = best = 0
    for i in range(max(0, d), min(len(q3), len(t3) + d)):
        run = max(0, run + S3(p, q3[i], t3[i - d]))
        best = max(best, run)
    return min(best, 255)


def test_sw_identity_known_answer(p):
    s = enc("ACDE")
    # (4+4) + (4+9) + (5+6) + (8+5) from the two matrices
    expect = sum(S3(p, c, c) + SA(p, c, c) for c in s)
    assert expect == 45
    assert O.sw(s, s, s, s, p) == (45, 3, 3)
    assert O.sw(s, s, s, s, p, rev_q=1, rev_t=1) == (45, 3, 3)


def test_sw_gap_known_answer(p):
    left, right = enc("WCHWCH"), enc("MFYMFY")
    q = np.concatenate([left, right])
    junk = enc("GG")
    t = np.concatenate([left, junk, right])
    full = sum(S3(p, c, c) + SA(p, c, c) for c in q)
    expect = full - (p.gap_open + (len(junk) - 1) * p.gap_ext)      # one gap of length 2: open + ext
    assert expect > full // 2 + 20
    assert O.sw(q, q, t, t, p) == (expect, len(q) - 1, len(t) - 1)
    assert O.sw(t, t, q, q, p) == (expect, len(t) - 1, len(q) - 1)   # gap on the other side


def test_sw_tiebreak_smallest_tend_then_qend(p):
    s = enc("WCHMFY")
    junk = enc("GGGG")
    twice = np.concatenate([s, junk, s])
    score = sum(S3(p, c, c) + SA(p, c, c) for c in s)
    # the same optimum ends in two target columns -> the first one
    assert O.sw(s, s, twice, twice, p) == (score, len(s) - 1, len(s) - 1)
    # ... and in two query rows of the same column -> the smaller row
    assert O.sw(twice, twice, s, s, p) == (score, len(s) - 1, len(s) - 1)


def test_sw_no_alignment(p):
    a, b = enc("DDDD"), enc("FFFF")     # S3[D][F] = -10, SA[D][F] = -3
    assert O.sw(a, a, b, b, p) == (0, -1, -1)


def test_sw_matches_pure_python_on_random_small_cases(p):
    rng = np.random.default_rng(7)
    for it in range(300):
        lq, lt = int(rng.integers(1, 24)), int(rng.integers(1, 24))
        q3, qa = rng.integers(0, 21, lq, dtype=np.uint8), rng.integers(0, 21, lq, dtype=np.uint8)
        if it % 3 == 0:   # related pair: makes gaps and ties likely
            keep = rng.random(lq) > 0.15
            t3, ta = q3[keep].copy(), qa[keep].copy()
            if len(t3) == 0:
                t3, ta = q3[:1].copy(), qa[:1].copy()
        else:
            t3, ta = rng.integers(0, 21, lt, dtype=np.uint8), rng.integers(0, 21, lt, dtype=np.uint8)
        assert O.sw(q3, qa, t3, ta, p) == py_sw(q3, qa, t3, ta, p)
        assert O.sw(q3, qa, t3, ta, p, rev_q=1) == py_sw(q3[::-1], qa[::-1], t3, ta, p)
        assert O.sw(q3, qa, t3, ta, p, rev_q=1, rev_t=1) == py_sw(q3[::-1], qa[::-1], t3[::-1], ta[::-1], p)


def test_ungapped_known_answers_and_python(p):
    s = enc("ACDEFGHIKL")
    self_score = sum(S3(p, c, c) for c in s)
    assert O.ungapped(s, s, 0, p) == self_score == 55
    assert O.ungapped(s, s, 100, p) == 0 and O.ungapped(s, s, -100, p) == 0    # no overlap
    long = np.tile(s, 10)
    assert O.ungapped(long, long, 0, p) == 255                                   # saturates like MMseqs2's uint8
    # Kadane reset: a strongly negative stretch must not leak into the second block
    q = np.concatenate([enc("EEEE"), enc("DDDD"), enc("MMMM")])
    t = np.concatenate([enc("EEEE"), enc("FFFF"), enc("MMMM")])
    assert O.ungapped(q, t, 0, p) == 4 * S3(p, 3, 3) == 32
    rng = np.random.default_rng(3)
    for _ in range(300):
        a, b = rng.integers(0, 21, int(rng.integers(1, 40)), dtype=np.uint8), rng.integers(0, 21, int(rng.integers(1, 40)), dtype=np.uint8)
        d = int(rng.integers(-45, 45))
        assert O.ungapped(a, b, d, p) == py_ungapped(a, b, d, p)


# ------------------------------------------------------------------ similar k-mers
def count_by_convolution(p, letters, thr):
    """number of 6-tuples with score sum >= thr via convolution of the per-position score histograms"""
    hist = {0: 1}
    for c in letters:
        nxt = {}
        for s, n in hist.items():
            for b in range(20):
                k = s + S3(p, c, b)
                nxt[k] = nxt.get(k, 0) + n
        hist = nxt
    return sum(n for s, n in hist.items() if s >= thr)


def test_similar_kmers_exact_set(p):
    rng = np.random.default_rng(1)
    for _ in range(20):
        letters = [int(x) for x in rng.integers(0, 20, 6)]
        self_score = sum(S3(p, c, c) for c in letters)
        for thr in (self_score + 1, self_score, self_score - 3, 24):
            got = O.similar_kmers(letters, thr, p)
            assert len(got) == len(set(got.tolist())) == count_by_convolution(p, letters, thr)
            for v in got[:200]:
                dec = [(int(v) // 20 ** m) % 20 for m in range(6)]
                assert sum(S3(p, letters[m], dec[m]) for m in range(6)) >= thr
        val = sum(c * 20 ** m for m, c in enumerate(letters))
        assert val in set(O.similar_kmers(letters, self_score, p).tolist())       # the k-mer itself
        assert len(O.similar_kmers(letters, self_score + 1, p)) == 0 or max(p.S3[c * 21 + b] for c in letters for b in range(20)) > 0


# ------------------------------------------------------------------ set cover
def py_setcover(n, edges):
    adj = [set() for _ in range(n)]
    for a, b in edges:
        if a != b:
            adj[a].add(b); adj[b].add(a)
    assign = [-1] * n
    while -1 in assign:
        best, bu = -1, -1
        for u in range(n):
            if assign[u] == -1:
                c = 1 + sum(1 for v in adj[u] if assign[v] == -1)
                if c > best:
                    best, bu = c, u
        assign[bu] = bu
        for v in adj[bu]:
            if assign[v] == -1:
                assign[v] = bu
    return assign


def test_setcover_known_answers():
    # star: centre 3 covers everything
    assert O.setcover(5, [(3, 0), (3, 1), (3, 2), (3, 4)]).tolist() == [3, 3, 3, 3, 3]
    # path 0-1-2-3-4: nodes 1,2,3 cover 3 each -> smallest id 1 first, then 3 covers {3,4}... (2 is taken by 1)
    assert O.setcover(5, [(0, 1), (1, 2), (2, 3), (3, 4)]).tolist() == [1, 1, 1, 3, 3]
    # no edges: singletons; duplicates / self loops / both directions are harmless
    assert O.setcover(3, np.zeros((0, 2), np.uint32)).tolist() == [0, 1, 2]
    assert O.setcover(3, [(0, 1), (1, 0), (0, 1), (2, 2)]).tolist() == [0, 0, 2]
    # two triangles joined by a bridge: tie on size 4 between 2 and 3 -> 2 first
    e = [(0, 1), (0, 2), (1, 2), (2, 3), (3, 4), (3, 5), (4, 5)]
    assert O.setcover(6, e).tolist() == [2, 2, 2, 2, 4, 4]


def test_setcover_matches_pure_python_random():
    rng = np.random.default_rng(5)
    for _ in range(60):
        n = int(rng.integers(1, 40))
        m = int(rng.integers(0, 3 * n))
        e = rng.integers(0, n, (m, 2))
        assert O.setcover(n, e).tolist() == py_setcover(n, e.tolist())


# ------------------------------------------------------------------ E-value gate, full pair, pipeline invariants
def test_min_score_is_the_smallest_passing_integer(p):
    import math
    for lq, res in [(50, 10 ** 4), (300, 45 * 10 ** 6), (2000, 18 * 10 ** 8)]:
        s = O.lib().uco_min_score(p, lq, res)
        ev = lambda x: p.K * lq * res * math.exp(-p.lambda_ * x)
        assert ev(s) <= p.evalue < ev(s - 1)


def test_align_pair_start_positions_and_coverage(p):
    core = enc("WCHMFYWCHMFYWCHMFYWCHMFY")                      # 24 strongly scoring residues
    q = np.concatenate([enc("DDD"), core, enc("DD")])            # core at 3..26 of 29; flanks mismatch hard (S3[D][F] = -10)
    t = np.concatenate([enc("FFFFF"), core])                     # core at 5..28 of 29
    odb = O.OracleDb(s3=[q, t], sa=[q, t])
    a = O.Aln()
    import ctypes
    O.lib().uco_align_pair(ctypes.byref(odb.db), 0, 1, ctypes.byref(p), 1, ctypes.byref(a))
    assert (a.qstart, a.qend, a.tstart, a.tend) == (3, 26, 5, 28)
    assert a.score == sum(S3(p, c, c) + SA(p, c, c) for c in core)
    assert a.pass_evalue == 1 and a.accepted == 1                 # 24/29 = 0.83 >= 0.8 on both sides
    p2 = O.default_params(cov=0.9)
    O.lib().uco_align_pair(ctypes.byref(odb.db), 0, 1, ctypes.byref(p2), 1, ctypes.byref(a))
    assert a.accepted == 0
    p3 = O.default_params(min_seq_id=0.5)                          # traceback path: all 24 columns identical
    O.lib().uco_align_pair(ctypes.byref(odb.db), 0, 1, ctypes.byref(p3), 1, ctypes.byref(a))
    assert (a.aln_len, a.idents, a.accepted) == (24, 24, 1)


def test_pipeline_on_family_db_recovers_families_and_tsv_invariants(tmp_path):
    s3, sa = util.family_db(21, n_fam=10, members=5, lmin=60, lmax=160, with_x=False)
    names = util.write_db(str(tmp_path / "db"), s3, sa)
    odb = O.OracleDb(str(tmp_path / "db"))
    assert odb.names() == names and odb.n == len(s3)
    got3, gota = odb.codes()
    assert np.array_equal(got3, np.concatenate(s3)) and np.array_equal(gota, np.concatenate(sa))
    p = util.oracle_params(O, "-c 0.8")
    r = O.cluster(odb, p, threads=4)
    a = r["assign"]
    for f in range(10):   # the 4 full-length members of every family share one representative
        assert len(set(a[f * 5: f * 5 + 4].tolist())) == 1
    assert len(set(a[:50].tolist())) >= 10
    assert (a[a] == a).all()          # representatives represent themselves
    O.write_tsv(str(tmp_path / "c.tsv"), odb, a)
    util.tsv_invariants(str(tmp_path / "c.tsv"), names)
    # every prefilter list is sorted by (score desc, target asc), holds the query itself when it has k-mers
    for q in range(odb.n):
        h = r["hits"][q, : r["hit_cnt"][q]]
        key = list(zip((-h["score"]).tolist(), h["t"].tolist()))
        assert key == sorted(key) and len(h) <= p.max_seqs
        if len(s3[q]) >= 30 and (s3[q] < 20).all():
            assert q in h["t"].tolist()


def test_cascade_oracle_invariants():
    """E8: one round == the single step; the merged assignment is idempotent (a representative represents
    itself), coarser than round 1, and every round runs on the representatives of the previous one"""
    import util
    from oracle import oracle_py as O
    s3, sa = util.family_db(5, n_fam=10, members=6, lmin=50, lmax=200)
    odb = O.OracleDb(s3=s3, sa=sa)
    p = util.oracle_params(O, "-c 0.8")
    one = O.cluster(odb, p, threads=4, dumps=False)
    c1 = O.cluster_cascade(odb, p, [p.kmer_thr], threads=4)
    assert np.array_equal(one["assign"], c1["assign"]) and c1["round_sizes"].tolist() == [odb.n]
    thr = O.cascade_thresholds(p, 4.0, 3)
    assert thr[-1] == p.kmer_thr and thr[0] > thr[1] > thr[2]
    c3 = O.cluster_cascade(odb, p, thr, threads=4)
    a = c3["assign"]
    assert all(a[a[i]] == a[i] for i in range(odb.n))
    rs = c3["round_sizes"].tolist()
    assert rs[0] == odb.n and rs[0] > rs[1] >= rs[2] >= c3["counts"]["n_clusters"] == len(set(a.tolist()))
    # round 1 alone (highest threshold) is a refinement of the merged result
    r1 = O.cluster_cascade(odb, p, thr[:1], threads=4)["assign"]
    assert all(a[r1[i]] == a[i] for i in range(odb.n))


def test_length_gate_rule(tmp_path):
    """optional rule UC-1/L (default off): MMseqs2's canBeCovered on the two lengths.  Known answers for the three coverage modes, off by
    default, and through the pipeline: gated pairs keep all-zero records and count neither as alignments nor in the cell counters, every
    other record and every prefilter list is the rule-off run's."""
    import util
    L = O.lib()
    L.uco_can_be_covered.argtypes = [C.c_void_p, C.c_int, C.c_int]
    p = util.oracle_params(O, "-c 0.8")
    assert p.len_gate == 0 and L.uco_can_be_covered(C.byref(p), 10, 1000) == 1            # off: everything passes
    for mode, cases in ((0, ((100, 100, 1), (80, 100, 1), (100, 80, 1), (79, 100, 0), (100, 79, 0), (4, 5, 1), (3, 5, 0))),
                        (1, ((80, 100, 1), (79, 100, 0), (500, 100, 1))),                 # target coverage: the query must be long enough
                        (2, ((100, 80, 1), (100, 79, 0), (100, 500, 1)))):                # query coverage: the target must be long enough
        p = util.oracle_params(O, "-c 0.8 --cov-mode %d --length-gate 1" % mode)
        for lq, lt, exp in cases:
            assert L.uco_can_be_covered(C.byref(p), lq, lt) == exp, (mode, lq, lt)
    assert L.uco_can_be_covered(C.byref(util.oracle_params(O, "-c 0 --length-gate 1")), 1, 1000) == 1   # no coverage threshold, no gate
    s3, sa = util.family_db(23, n_fam=10, members=6, lmin=30, lmax=300, indel=0.08, extra=(500,))
    odb = O.OracleDb(s3=s3, sa=sa)
    on = O.cluster(odb, util.oracle_params(O, "-c 0.8 --length-gate 1"), threads=4)
    off = O.cluster(odb, util.oracle_params(O, "-c 0.8"), threads=4)
    assert np.array_equal(on["hit_cnt"], off["hit_cnt"]) and on["hits"].tobytes() == off["hits"].tobytes()      # E1-E4 untouched
    lens = np.array([len(x) for x in s3], np.float32)
    gated = kept = 0
    for q in range(odb.n):
        for k in range(on["hit_cnt"][q]):
            t = int(on["hits"][q, k]["t"])
            if min(lens[q] / lens[t], lens[t] / lens[q]) >= np.float32(0.8):
                assert on["aln"][q, k].tobytes() == off["aln"][q, k].tobytes(); kept += 1
            else:
                assert not np.frombuffer(on["aln"][q, k].tobytes(), np.uint8).any(); gated += 1
    assert gated > 20 and kept > 20 and on["counts"]["n_alignments"] == kept and off["counts"]["n_alignments"] == kept + gated
    assert on["counts"]["cells_fwd"] < off["counts"]["cells_fwd"] and on["counts"]["n_edges"] <= off["counts"]["n_edges"]


@pytest.mark.parametrize("opts", ["-c 0.8", "-c 0.5 -e 1e-2 --min-seq-id 0.3 --gap-open 7 --gap-extend 2 --cov-mode 1", "-c 0.8 --rev-correction 0 --max-seqs 9",
                                  "-c 0.8 --length-gate 1", "-c 0.7 --cov-mode 2 --length-gate 1 --min-seq-id 0.3"])
def test_simd_cpu_leg_equals_the_scalar_oracle(opts, tmp_path):
    """oracle/uc_simd.c (bench.py's cpu_baseline "simd": AVX2 inter-sequence Smith-Waterman, 16 targets per register)
    must give the scalar oracle's record for every pair - score, reversed-query score, ends, starts, gates, traceback
    statistics - including sequences of very different lengths in one batch, X residues and 1-residue sequences."""
    import util
    s3, sa = util.family_db(11, n_fam=14, members=7, lmin=12, lmax=420, extra=(700, 33, 1500))
    db = str(tmp_path / "db")
    util.write_db(db, s3, sa)
    odb = O.OracleDb(db)
    p = util.oracle_params(O, opts)
    ix = O.build_index(odb, p)
    queries = np.arange(odb.n, dtype=np.uint32)[::-1].copy()
    n, _, _, cnt, hits, alns = O.simd_sample_run(odb, ix, p, queries, threads=4, records=True)
    n_ref, _, _ = O.sample_run(odb, ix, p, queries, threads=4)
    O.free_index(ix)
    assert n == n_ref and n > 200 and (n < int(cnt.sum()) if "--length-gate 1" in opts else n == int(cnt.sum()))   # gated pairs are no alignments
    checked = 0
    for k, q in enumerate(queries):
        ms = O.min_score(odb, p, int(q))
        for h in range(cnt[k]):
            ref = O.align_pair(odb, p, int(q), int(hits[k, h]["t"]), ms)
            got = alns[k, h]
            for f in ("score", "score_rev", "corrected", "pass_evalue", "accepted", "aln_len", "idents", "gap_opens"):
                assert got[f] == ref[f], (opts, int(q), int(hits[k, h]["t"]), f, got, ref)
            if ref["pass_evalue"]:
                for f in ("qstart", "qend", "tstart", "tend"):
                    assert got[f] == ref[f], (opts, int(q), int(hits[k, h]["t"]), f, got, ref)
            checked += 1
    assert checked == int(cnt.sum())


# ------------------------------------------------------------------ published constants / textbook properties
def test_blosum62_is_the_ncbi_matrix(p):
    """BLOSUM62 (Henikoff & Henikoff 1992, NCBI half-bit rounding): the diagonal, symmetry, range and a column are fixed
    published values that do not depend on any Foldseek data"""
    idx = {c: i for i, c in enumerate(LET)}
    diag = dict(A=4, R=5, N=6, D=6, C=9, Q=5, E=5, G=6, H=8, I=4, L=4, K=5, M=5, F=6, P=7, S=4, T=5, W=11, Y=7, V=4)
    m = np.array(p.SA[:]).reshape(21, 21)[:20, :20]
    assert all(m[idx[c], idx[c]] == v for c, v in diag.items()) and int(np.trace(m)) == 116
    assert (m == m.T).all() and m.min() == -4 and m.max() == 11
    w_col = dict(A=-3, R=-3, N=-4, D=-4, C=-2, Q=-2, E=-3, G=-2, H=-2, I=-3, L=-2, K=-3, M=-1, F=1, P=-4, S=-3, T=-2, W=11, Y=2, V=-3)
    assert all(m[idx[c], idx["W"]] == v for c, v in w_col.items())
    # expected score of BLOSUM62 under the Robinson & Robinson background is negative (a local-alignment matrix must be)
    bg = dict(A=.078, R=.051, N=.045, D=.054, C=.019, Q=.043, E=.063, G=.074, H=.022, I=.051, L=.091, K=.057, M=.022, F=.039, P=.052, S=.071, T=.058, W=.013, Y=.032, V=.064)
    f = np.array([bg[c] for c in LET])
    assert -1.2 < float(f @ m @ f) < -0.7


def test_spaced_seed_is_the_mmseqs2_k6_pattern(p):
    """MMseqs2's spaced seed for k = 6 (src/commons/Sequence.h: seed_6_spaced = {1,1,0,1,0,1,0,0,1,1}; EXT-UNVERIFIED for
    Foldseek, which embeds MMseqs2): span 10, offsets 0 1 3 5 8 9 - the default of both the oracle and the engine"""
    off = (C.c_int * 6)()
    assert O.lib().uco_pattern_offsets(b"1101010011", off) == 10 and list(off) == [0, 1, 3, 5, 8, 9]
    assert (p.pattern.decode() if isinstance(p.pattern, bytes) else p.pattern) == "1101010011"
    import unicore_amd as U
    assert U.check_options("--spaced-kmer-pattern 1101010011") == 0


def test_affine_gap_properties_of_gotoh(p):
    """Gotoh 1982 affine gaps as the spec fixes them (a gap of k residues costs open + (k - 1) * ext): one long gap beats
    two short ones exactly when it is cheaper, gaps in query and target are symmetric, and a gap is never opened when the
    mismatch path scores more"""
    blk = [enc(x) for x in ("WCHWCHWC", "MFYMFYMF", "HWCKPRHW")]
    a, b, c = blk
    junk1, junk3 = enc("G"), enc("GGG")
    full = lambda s: sum(S3(p, x, x) + SA(p, x, x) for x in s)
    q = np.concatenate([a, b, c])
    # two separate 1-residue insertions in the target: 2 * open
    t2 = np.concatenate([a, junk1, b, junk1, c])
    assert O.sw(q, q, t2, t2, p)[0] == full(q) - 2 * p.gap_open
    # one 3-residue insertion: open + 2 ext (cheaper than the two above although it skips more residues)
    t1 = np.concatenate([a, junk3, b, c])
    assert O.sw(q, q, t1, t1, p)[0] == full(q) - (p.gap_open + 2 * p.gap_ext)
    assert O.sw(t1, t1, q, q, p)[0] == O.sw(q, q, t1, t1, p)[0]                 # deletion == insertion
    # a single substituted residue is aligned as a mismatch, not bridged by two gaps
    m = q.copy(); m[10] = enc("G")[0]
    mm = S3(p, q[10], m[10]) + SA(p, q[10], m[10]) - (S3(p, q[10], q[10]) + SA(p, q[10], q[10]))
    assert mm > -2 * p.gap_open and O.sw(q, q, m, m, p)[0] == full(q) + mm


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """the C checker (scalar + SIMD legs) through gcc's address and undefined-behaviour sanitizers on a small family DB:
    prefilter, gapped passes, traceback, cascade + pre-step, set cover, TSV writer - no report, same result as the normal build"""
    import subprocess, sys
    so = os.path.join(ROOT, "oracle", "_asan", "liboracle_asan.so")
    subprocess.check_call(["make", "-C", ROOT, "oracle-asan"], stdout=subprocess.DEVNULL)
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import util
from oracle import oracle_py as O
s3, sa = util.family_db(5, n_fam=6, members=5, extra=(300,))
odb = O.OracleDb(s3=s3, sa=sa)
p = util.oracle_params(O, "-c 0.5 --min-seq-id 0.3")
r = O.cluster(odb, p, threads=2)
w = O.cluster_workflow(odb, p, O.cascade_thresholds(p, 4.0, 2), linclust_m=5, threads=2)
ix = O.build_index(odb, p)
n, _, _, cnt, hits, alns = O.simd_sample_run(odb, ix, p, np.arange(odb.n, dtype=np.uint32), threads=2, records=True)
O.free_index(ix)
O.write_tsv(%r, odb, r["assign"])
pg = util.oracle_params(O, "-c 0.8 --length-gate 1 --min-seq-id 0.3")      # optional rule UC-1/L through both legs
rg = O.cluster(odb, pg, threads=2)
ix = O.build_index(odb, pg)
ng, _, _, _, _, ag = O.simd_sample_run(odb, ix, pg, np.arange(odb.n, dtype=np.uint32), threads=2, records=True)
O.free_index(ix)
print("RESULT", int(r["counts"]["n_alignments"]), int(r["counts"]["n_clusters"]), int(w["counts"]["n_clusters"]), n, int(alns["accepted"].sum()),
      int(rg["counts"]["n_alignments"]), int(rg["counts"]["n_clusters"]), ng, int(ag["accepted"].sum()))
""" % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path / "a.tsv"))
    outs = []
    for lib_env in ({"UC_ORACLE_LIB": so, "LD_PRELOAD": subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip(),
                     "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1"}, {}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **lib_env), timeout=600)
        assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0])
    assert outs[0] == outs[1]


def test_foldseek_diff_tool_self_test():
    """tools/foldseek_diff.py — the per-stage differ for the day a real `foldseek` binary is on PATH — parses MMseqs-style
    result databases and names the first diverging stage: its self-test writes the oracle's dumps as such databases, finds no difference,
    then finds one planted difference per stage (prefilter, alignment, set cover)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "foldseek_diff.py"), "--self-test"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "self-test ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "foldseek_diff.py"), "/nonexistent/db", "--foldseek", os.path.join(ROOT, "bin", "foldseek")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no real foldseek" in (r.stdout + r.stderr)      # this repository's shim is never mistaken for Foldseek


def test_matrix_bit_factor_rule_is_the_same_on_both_sides():
    """optional rule UC-1/M (default off): matrices rescaled to `bit factor` units per bit from the file's lambda (half-bit units without a
    header): the oracle's C restatement and the engine's loader give the same integers, and the engine rejects what the oracle would"""
    import unicore_amd as U
    p0 = util.oracle_params(O, "-c 0.8")
    p1 = util.oracle_params(O, "-c 0.8 --mat-bit-factor-3di 2.1 --mat-bit-factor-aa 1.4")
    s0, s1 = np.array(p0.S3[:], np.int64), np.array(p1.S3[:], np.int64)
    a0, a1 = np.array(p0.SA[:], np.int64), np.array(p1.SA[:], np.int64)
    assert np.array_equal(s1, np.sign(s0) * np.floor(np.abs(s0) * 1.05 + 0.5).astype(np.int64))      # lround: half away from zero
    assert np.array_equal(a1, np.sign(a0) * np.floor(np.abs(a0) * 0.7 + 0.5).astype(np.int64))
    assert U.check_options("-c 0.8 --mat-bit-factor-3di 2.1 --mat-bit-factor-aa 1.4") == 0
    assert U.check_options("-c 0.8 --mat-bit-factor-3di 40") != 0
    L = O.lib()
    import ctypes as C
    L.uco_matrix_header_lambda.argtypes = [C.c_char_p]
    L.uco_matrix_header_lambda.restype = C.c_double
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".out", delete=False) as f:
        f.write("# 3Di bit/2\n# Background (precision=3):\n# 0.05 0.05\n# Lambda     (precision=3):\n# 0.351568\n   A   C\nA   6  -3\nC  -3   9\n")
    assert abs(L.uco_matrix_header_lambda(f.name.encode()) - 0.351568) < 1e-9
    assert L.uco_matrix_header_lambda(O.data_path("blosum62.out").encode()) == 0.0
    os.unlink(f.name)


def test_compositional_bias_rule_known_answers():
    """optional rule UC-1/B: bias_i = round_half_away(scale * (rowsum(q_i) / 20 - sum_{j in window, j != i} S[q_i][q_j] / |window|)), window =
    [max(0, i - 20), min(L, i + 20)) — checked against a direct float64 restatement (ties excluded by construction: exact integer arithmetic in C)
    and against a hand-computed case; the biased ungapped score adds bias_i per QUERY position"""
    import ctypes as C
    L = O.lib()
    p = O.default_params()
    S = np.array(p.S3[:], np.int32).reshape(21, 21)
    rng = np.random.default_rng(3)
    L.uco_comp_bias.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.uco_ungapped_bias.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.uco_ungapped_bias.restype = C.c_int32
    for n, scale in ((1, 1000), (7, 1000), (45, 500), (130, 1000), (130, 2500)):
        q = rng.integers(0, 21, n).astype(np.uint8)
        out = np.zeros(n, np.int8)
        L.uco_comp_bias(q.ctypes.data, n, C.addressof(p.S3), scale, out.ctypes.data)
        for i in range(n):
            lo, hi = max(0, i - 20), min(n, i + 20)
            sm = int(S[q[i], q[lo:hi]].sum() - S[q[i], q[i]])
            from fractions import Fraction
            v = Fraction(scale, 1000) * (Fraction(int(S[q[i], :20].sum()), 20) - Fraction(sm, hi - lo))
            exp = int(v + Fraction(1, 2)) if v >= 0 else -int(-v + Fraction(1, 2))
            assert out[i] == max(-128, min(127, exp)), (n, scale, i)
    # a sequence of one repeated letter: every neighbour scores the diagonal entry -> strongly negative bias (low complexity is penalised)
    q = np.full(50, 3, np.uint8)
    out = np.zeros(50, np.int8)
    L.uco_comp_bias(q.ctypes.data, 50, C.addressof(p.S3), 1000, out.ctypes.data)
    assert (out < 0).all() and out[25] == round(float(S[3, :20].sum()) / 20 - float(S[3, 3]) * 39 / 40)
    # ungapped with bias = Kadane over S[q_i][t_j] + bias_i
    t = rng.integers(0, 20, 60).astype(np.uint8)
    qq = rng.integers(0, 20, 50).astype(np.uint8)
    L.uco_comp_bias(qq.ctypes.data, 50, C.addressof(p.S3), 1000, out.ctypes.data)
    for d in (-7, 0, 5):
        run = best = 0
        for i in range(max(d, 0), min(50, 60 + d)):
            run = max(0, run + int(S[qq[i], t[i - d]]) + int(out[i])); best = max(best, run)
        assert L.uco_ungapped_bias(qq.ctypes.data, 50, t.ctypes.data, 60, d, C.addressof(p.S3), out.ctypes.data) == min(best, 255)
