"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product package
(unicore_amd) never imports this module.  See oracle/uc_oracle.h for the frozen spec UC-1 and the
"parity unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
A = 21
K = 6


class Params(C.Structure):
    _fields_ = [
        ("S3", C.c_int8 * (A * A)), ("SA", C.c_int8 * (A * A)), ("pattern", C.c_char * 33),
        ("kmer_thr", C.c_int), ("min_diag_hits", C.c_int), ("min_ungapped", C.c_int), ("max_seqs", C.c_int),
        ("gap_open", C.c_int), ("gap_ext", C.c_int), ("rev_correction", C.c_int),
        ("evalue", C.c_double), ("lambda_", C.c_double), ("K", C.c_double),
        ("cov", C.c_float), ("cov_mode", C.c_int), ("min_seq_id", C.c_float), ("want_tb", C.c_int),
        ("comp_bias_milli", C.c_int), ("min_score_table", C.c_void_p),       # optional rules UC-1/B, UC-1/E (default off)
        ("len_gate", C.c_int),                                               # optional rule UC-1/L (default off)
    ]


class Db(C.Structure):
    _fields_ = [("n", C.c_uint32), ("off", C.POINTER(C.c_uint64)), ("s3", C.POINTER(C.c_uint8)),
                ("sa", C.POINTER(C.c_uint8)), ("names", C.POINTER(C.c_char_p))]


class Index(C.Structure):
    _fields_ = [("koff", C.POINTER(C.c_uint32)), ("ent_seq", C.POINTER(C.c_uint32)),
                ("ent_pos", C.POINTER(C.c_uint16)), ("n_entries", C.c_uint64)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_uint32), ("score", C.c_int32), ("diag", C.c_int32)]


class Aln(C.Structure):
    _fields_ = [("score", C.c_int32), ("score_rev", C.c_int32), ("corrected", C.c_int32),
                ("qstart", C.c_int32), ("qend", C.c_int32), ("tstart", C.c_int32), ("tend", C.c_int32),
                ("aln_len", C.c_int32), ("idents", C.c_int32), ("pass_evalue", C.c_int32), ("accepted", C.c_int32),
                ("gap_opens", C.c_int32)]


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_sim_kmers", "n_kmer_hits", "n_candidates", "n_prefilter_hits",
                                          "n_alignments", "n_edges", "n_clusters", "cells_fwd", "cells_rev", "cells_start")]


HIT_DTYPE = np.dtype([("t", "<u4"), ("score", "<i4"), ("diag", "<i4")])
ALN_DTYPE = np.dtype([(n, "<i4") for n in ("score", "score_rev", "corrected", "qstart", "qend", "tstart", "tend",
                                            "aln_len", "idents", "pass_evalue", "accepted", "gap_opens")])

_lib = None


def lib():
    """Load (building if necessary — gcc is part of the image) oracle/liboracle.so."""
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "uc_oracle.c")
    if os.environ.get("UC_ORACLE_LIB"):          # e.g. the address/UB-sanitizer build the CPU suite runs once (Makefile: oracle-asan)
        so = os.environ["UC_ORACLE_LIB"]
    elif not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ROOT, "oracle"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.uco_letter_code.argtypes = [C.c_char]
    L.uco_params_default.argtypes = [C.POINTER(Params)]
    L.uco_load_matrix.argtypes = [C.c_char_p, C.POINTER(C.c_int8)]
    L.uco_db_read.argtypes = [C.c_char_p, C.POINTER(Db)]
    L.uco_db_free.argtypes = [C.POINTER(Db)]
    L.uco_index_build.argtypes = [C.POINTER(Db), C.c_uint32, C.c_uint32, C.POINTER(Params), C.POINTER(Index)]
    L.uco_index_free.argtypes = [C.POINTER(Index)]
    L.uco_similar_kmers.argtypes = [C.POINTER(C.c_int8), C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint32), C.c_size_t]
    L.uco_similar_kmers.restype = C.c_size_t
    L.uco_ungapped.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int8)]
    L.uco_ungapped.restype = C.c_int32
    L.uco_prefilter_query.argtypes = [C.POINTER(Db), C.POINTER(Index), C.c_uint32, C.POINTER(Params), C.c_void_p, C.POINTER(Counts)]
    L.uco_sw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                         C.POINTER(Params), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.uco_min_score.argtypes = [C.POINTER(Params), C.c_int, C.c_uint64]
    L.uco_min_score.restype = C.c_int32
    L.uco_align_pair.argtypes = [C.POINTER(Db), C.c_uint32, C.c_uint32, C.POINTER(Params), C.c_int32, C.POINTER(Aln)]
    L.uco_setcover.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    L.uco_cluster.argtypes = [C.POINTER(Db), C.POINTER(Params), C.c_int, C.c_void_p, C.POINTER(Counts), C.c_void_p, C.c_void_p, C.c_void_p]
    L.uco_cluster_cascade.argtypes = [C.POINTER(Db), C.POINTER(Params), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.POINTER(Counts), C.c_void_p]
    L.uco_write_tsv.argtypes = [C.c_char_p, C.POINTER(Db), C.c_void_p]
    L.uco_linclust_pairs.argtypes = [C.POINTER(Db), C.POINTER(Params), C.c_int, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64)]
    L.uco_cluster_workflow.argtypes = [C.POINTER(Db), C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.POINTER(Counts), C.c_void_p]
    L.uco_search.argtypes = [C.POINTER(Db), C.POINTER(Db), C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Counts)]
    L.uco_write_m8.argtypes = [C.c_char_p, C.POINTER(Db), C.POINTER(Db), C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p]
    L.uco_sample_run.argtypes = [C.POINTER(Db), C.POINTER(Index), C.POINTER(Params), C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
    L.uco_sample_run.restype = C.c_uint64
    L.uco_simd_sample_run.argtypes = [C.POINTER(Db), C.POINTER(Index), C.POINTER(Params), C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_double),
                                      C.c_void_p, C.c_void_p, C.c_void_p]
    L.uco_simd_sample_run.restype = C.c_uint64
    L.uco_simd_sample_run_counts.argtypes = L.uco_simd_sample_run.argtypes + [C.POINTER(Counts)]
    L.uco_simd_sample_run_counts.restype = C.c_uint64
    L.uco_align_pair.argtypes = [C.POINTER(Db), C.c_uint32, C.c_uint32, C.POINTER(Params), C.c_int32, C.c_void_p]
    L.uco_align_pair.restype = None
    L.uco_min_score.argtypes = [C.POINTER(Params), C.c_int, C.c_uint64]
    L.uco_min_score.restype = C.c_int32
    _lib = L
    return L


def data_path(name):
    return os.path.join(_ROOT, "unicore_amd", "data", name)


def default_params(**over):
    p = Params()
    lib().uco_params_default(C.byref(p))
    assert lib().uco_load_matrix(data_path("mat3di_synthetic.out").encode(), p.S3) == 0
    assert lib().uco_load_matrix(data_path("blosum62.out").encode(), p.SA) == 0
    bf3, bfa = over.pop("bit_factor_3di", 0.0), over.pop("bit_factor_aa", 0.0)
    L = lib()
    L.uco_rescale_matrix.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.uco_matrix_header_lambda.argtypes = [C.c_char_p]
    L.uco_matrix_header_lambda.restype = C.c_double
    if bf3 > 0:
        assert L.uco_rescale_matrix(C.addressof(p.S3), bf3, L.uco_matrix_header_lambda(data_path("mat3di_synthetic.out").encode())) == 0
    if bfa > 0:
        assert L.uco_rescale_matrix(C.addressof(p.SA), bfa, L.uco_matrix_header_lambda(data_path("blosum62.out").encode())) == 0
    for k, v in over.items():
        if k == "pattern":
            v = v.encode() if isinstance(v, str) else v
        setattr(p, "lambda_" if k == "lambda" else k, v)
    return p


def encode(seq):
    """letters -> uint8 codes (same table as uco_letter_code)."""
    lut = np.full(256, 20, np.uint8)
    for i, c in enumerate("ACDEFGHIKLMNPQRSTVWY"):
        lut[ord(c)] = i
        lut[ord(c.lower())] = i
    return lut[np.frombuffer(seq.encode() if isinstance(seq, str) else seq, np.uint8)]


class OracleDb:
    """In-memory DB for the oracle, either read from disk or built from code arrays."""

    def __init__(self, prefix=None, s3=None, sa=None, names=None):
        self.db = Db()
        self._keep = []
        if prefix is not None:
            rc = lib().uco_db_read(prefix.encode(), C.byref(self.db))
            if rc != 0:
                raise IOError("uco_db_read(%s) failed: %d" % (prefix, rc))
            self._owned = True
        else:
            n = len(s3)
            lens = np.array([len(x) for x in s3], np.uint64)
            off = np.zeros(n + 1, np.uint64)
            off[1:] = np.cumsum(lens)
            cat3 = np.concatenate([np.asarray(x, np.uint8) for x in s3] + [np.zeros(1, np.uint8)])
            cata = np.concatenate([np.asarray(x, np.uint8) for x in sa] + [np.zeros(1, np.uint8)])
            names = names or ["seq%d" % i for i in range(n)]
            arr = (C.c_char_p * n)(*[x.encode() for x in names])
            self._keep = [off, cat3, cata, arr]
            self.db.n = n
            self.db.off = off.ctypes.data_as(C.POINTER(C.c_uint64))
            self.db.s3 = cat3.ctypes.data_as(C.POINTER(C.c_uint8))
            self.db.sa = cata.ctypes.data_as(C.POINTER(C.c_uint8))
            self.db.names = C.cast(arr, C.POINTER(C.c_char_p))
            self._owned = False

    @classmethod
    def from_flat(cls, off, c3, ca):
        """from concatenated code arrays + n+1 offsets (no names: enough for everything but the TSV / m8 writers)"""
        self = cls.__new__(cls)
        self.db = Db()
        off = np.ascontiguousarray(off, np.uint64)
        tot = int(off[-1])
        cat3 = np.concatenate([np.asarray(c3[:tot], np.uint8), np.zeros(1, np.uint8)])
        cata = np.concatenate([np.asarray(ca[:tot], np.uint8), np.zeros(1, np.uint8)])
        self._keep = [off, cat3, cata]
        self.db.n = len(off) - 1
        self.db.off = off.ctypes.data_as(C.POINTER(C.c_uint64))
        self.db.s3 = cat3.ctypes.data_as(C.POINTER(C.c_uint8))
        self.db.sa = cata.ctypes.data_as(C.POINTER(C.c_uint8))
        self.db.names = None
        self._owned = False
        return self

    def subset(self, ids):
        """the sub-database of sequences `ids` (in that order): what a cascade round works on"""
        ids = np.asarray(ids, np.int64)
        off = self.offsets().astype(np.int64)
        if len(ids) == self.n and np.array_equal(ids, np.arange(self.n)):
            c3, ca = self.codes()
            return OracleDb.from_flat(off, c3, ca)
        lens = off[ids + 1] - off[ids]
        noff = np.zeros(len(ids) + 1, np.int64)
        noff[1:] = np.cumsum(lens)
        idx = np.repeat(off[ids] - noff[:-1], lens) + np.arange(int(noff[-1]), dtype=np.int64)
        tot = int(off[-1])
        s3 = np.ctypeslib.as_array(self.db.s3, shape=(tot,))
        sa = np.ctypeslib.as_array(self.db.sa, shape=(tot,))
        return OracleDb.from_flat(noff, s3[idx], sa[idx])

    @property
    def n(self):
        return int(self.db.n)

    def offsets(self):
        return np.ctypeslib.as_array(self.db.off, shape=(self.n + 1,)).copy()

    def codes(self):
        tot = int(self.offsets()[-1])
        return (np.ctypeslib.as_array(self.db.s3, shape=(tot,)).copy(),
                np.ctypeslib.as_array(self.db.sa, shape=(tot,)).copy())

    def names(self):
        return [self.db.names[i].decode() for i in range(self.n)]

    def __del__(self):
        if getattr(self, "_owned", False) and lib is not None and C is not None:   # module globals are gone at interpreter exit
            lib().uco_db_free(C.byref(self.db))


def sw(q3, qa, t3, ta, p, rev_q=0, rev_t=0):
    q3 = np.ascontiguousarray(q3, np.uint8); qa = np.ascontiguousarray(qa, np.uint8)
    t3 = np.ascontiguousarray(t3, np.uint8); ta = np.ascontiguousarray(ta, np.uint8)
    s, qe, te = C.c_int32(), C.c_int32(), C.c_int32()
    lib().uco_sw(q3.ctypes.data, qa.ctypes.data, len(q3), rev_q, t3.ctypes.data, ta.ctypes.data, len(t3), rev_t,
                 C.byref(p), C.byref(s), C.byref(qe), C.byref(te))
    return s.value, qe.value, te.value


def ungapped(q3, t3, diag, p):
    q3 = np.ascontiguousarray(q3, np.uint8); t3 = np.ascontiguousarray(t3, np.uint8)
    return int(lib().uco_ungapped(q3.ctypes.data, len(q3), t3.ctypes.data, len(t3), diag, p.S3))


def similar_kmers(letters, thr, p):
    c = (C.c_uint8 * K)(*letters)
    n = lib().uco_similar_kmers(p.S3, c, thr, None, 0)
    out = np.zeros(max(n, 1), np.uint32)
    lib().uco_similar_kmers(p.S3, c, thr, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    return out[:n]


def setcover(n, edges):
    e = np.ascontiguousarray(edges, np.uint32).reshape(-1, 2)
    assign = np.zeros(n, np.uint32)
    lib().uco_setcover(n, e.ctypes.data, len(e), assign.ctypes.data)
    return assign


def cluster(odb, p, threads=0, dumps=True):
    """Full pipeline.  Returns dict(assign, counts, hits, hit_cnt, aln)."""
    n, M = odb.n, p.max_seqs
    assign = np.zeros(n, np.uint32)
    cnt = Counts()
    hits = np.zeros((n, M), HIT_DTYPE) if dumps else None
    hcnt = np.zeros(n, np.uint32) if dumps else None
    aln = np.zeros((n, M), ALN_DTYPE) if dumps else None
    rc = lib().uco_cluster(C.byref(odb.db), C.byref(p), threads, assign.ctypes.data, C.byref(cnt),
                           hits.ctypes.data if dumps else None, hcnt.ctypes.data if dumps else None,
                           aln.ctypes.data if dumps else None)
    if rc != 0:
        raise RuntimeError("uco_cluster failed: %d" % rc)
    return dict(assign=assign, counts={f: getattr(cnt, f) for f, _ in Counts._fields_}, hits=hits, hit_cnt=hcnt, aln=aln)


def cascade_thresholds(p_single, sensitivity, steps, base=None):
    """k-mer threshold per round: sensitivity rises linearly from 1 to the target; same s -> threshold rule as
    the single step (round(6 * mean diag + 3 - 2 s), see uc_options.cpp / tests/util.py)."""
    diag = np.mean([p_single.S3[a * 21 + a] for a in range(20)])
    out = []
    for r in range(steps):
        s = sensitivity if steps == 1 else 1.0 + (sensitivity - 1.0) * r / (steps - 1)
        out.append(int(np.floor(6 * diag + 3.0 - 2.0 * s + 0.5)))
    return out


def cluster_cascade(odb, p, thr, threads=0):
    """E8: rounds on representatives + merge.  Returns dict(assign, counts, round_sizes)."""
    n, steps = odb.n, len(thr)
    assign = np.zeros(n, np.uint32)
    cnt = Counts()
    rs = np.zeros(steps, np.uint32)
    t = (C.c_int * steps)(*thr)
    rc = lib().uco_cluster_cascade(C.byref(odb.db), C.byref(p), steps, t, threads, assign.ctypes.data, C.byref(cnt), rs.ctypes.data)
    if rc != 0:
        raise RuntimeError("uco_cluster_cascade failed: %d" % rc)
    return dict(assign=assign, counts={f: getattr(cnt, f) for f, _ in Counts._fields_}, round_sizes=rs)


def search(qdb, tdb, p, threads=0):
    """search path: every query of qdb against tdb.  Returns dict(hits, hit_cnt, aln, counts) (nq x max_seqs)."""
    nq, M = qdb.n, p.max_seqs
    hits = np.zeros((nq, M), HIT_DTYPE)
    hcnt = np.zeros(nq, np.uint32)
    aln = np.zeros((nq, M), ALN_DTYPE)
    cnt = Counts()
    rc = lib().uco_search(C.byref(qdb.db), C.byref(tdb.db), C.byref(p), threads, hits.ctypes.data, hcnt.ctypes.data, aln.ctypes.data, C.byref(cnt))
    if rc != 0:
        raise RuntimeError("uco_search failed: %d" % rc)
    return dict(hits=hits, hit_cnt=hcnt, aln=aln, counts={f: getattr(cnt, f) for f, _ in Counts._fields_})


def write_m8(path, qdb, tdb, p, res):
    if lib().uco_write_m8(path.encode(), C.byref(qdb.db), C.byref(tdb.db), C.byref(p), res["hits"].ctypes.data,
                          res["hit_cnt"].ctypes.data, res["aln"].ctypes.data) != 0:
        raise IOError(path)


def linclust_pairs(odb, p, m=20):
    """E8a candidate pairs (centre, member), sorted and unique."""
    ptr = C.POINTER(C.c_uint32)()
    n = C.c_uint64()
    if lib().uco_linclust_pairs(C.byref(odb.db), C.byref(p), m, C.byref(ptr), C.byref(n)) != 0:
        raise RuntimeError("uco_linclust_pairs failed")
    out = np.ctypeslib.as_array(ptr, shape=(max(n.value, 1), 2))[: n.value].copy()
    C.CDLL(None).free(ptr)
    return out


def cluster_workflow(odb, p, thr, linclust_m=0, threads=0):
    """optional E8a pre-step + len(thr) cascade rounds.  Returns dict(assign, counts, round_sizes)."""
    n, steps = odb.n, len(thr)
    assign = np.zeros(n, np.uint32)
    cnt = Counts()
    rs = np.zeros(steps + 1, np.uint32)
    t = (C.c_int * max(steps, 1))(*thr)
    rc = lib().uco_cluster_workflow(C.byref(odb.db), C.byref(p), linclust_m, steps, t, threads, assign.ctypes.data, C.byref(cnt), rs.ctypes.data)
    if rc != 0:
        raise RuntimeError("uco_cluster_workflow failed: %d" % rc)
    return dict(assign=assign, counts={f: getattr(cnt, f) for f, _ in Counts._fields_}, round_sizes=rs[: steps + (1 if linclust_m > 0 else 0)])


def write_tsv(path, odb, assign):
    a = np.ascontiguousarray(assign, np.uint32)
    if lib().uco_write_tsv(path.encode(), C.byref(odb.db), a.ctypes.data) != 0:
        raise IOError(path)


def build_index(odb, p, tbegin=0, tend=None):
    ix = Index()
    rc = lib().uco_index_build(C.byref(odb.db), tbegin, odb.n if tend is None else tend, C.byref(p), C.byref(ix))
    if rc != 0:
        raise RuntimeError("uco_index_build failed: %d" % rc)
    return ix


def free_index(ix):
    lib().uco_index_free(C.byref(ix))


def sample_run(odb, ix, p, queries, threads=0):
    """E2-E6 for `queries` on the CPU; returns (n_alignments, prefilter_seconds, align_seconds)."""
    q = np.ascontiguousarray(queries, np.uint32)
    sec = (C.c_double * 2)()
    n = lib().uco_sample_run(C.byref(odb.db), C.byref(ix), C.byref(p), threads, q.ctypes.data, len(q), sec)
    return int(n), sec[0], sec[1]


def simd_sample_run(odb, ix, p, queries, threads=0, records=False):
    """sample_run with the gapped stage as AVX2 inter-sequence SW (oracle/uc_simd.c).  records=True also returns
    (counts, hits[nq, max_seqs], alns[nq, max_seqs])."""
    q = np.ascontiguousarray(queries, np.uint32)
    sec = (C.c_double * 2)()
    if not records:
        n = lib().uco_simd_sample_run(C.byref(odb.db), C.byref(ix), C.byref(p), threads, q.ctypes.data, len(q), sec, None, None, None)
        return int(n), sec[0], sec[1]
    M = p.max_seqs
    cnt, hits, alns = np.zeros(len(q), np.uint32), np.zeros((len(q), M), HIT_DTYPE), np.zeros((len(q), M), ALN_DTYPE)
    n = lib().uco_simd_sample_run(C.byref(odb.db), C.byref(ix), C.byref(p), threads, q.ctypes.data, len(q), sec,
                                  hits.ctypes.data, cnt.ctypes.data, alns.ctypes.data)
    return int(n), sec[0], sec[1], cnt, hits, alns


def simd_run_counts(odb, ix, p, queries, threads=0):
    """simd_sample_run(records=True) plus the prefilter's stage counters of these queries (dict)"""
    q = np.ascontiguousarray(queries, np.uint32)
    sec = (C.c_double * 2)()
    M = p.max_seqs
    cnt, hits, alns = np.zeros(len(q), np.uint32), np.zeros((len(q), M), HIT_DTYPE), np.zeros((len(q), M), ALN_DTYPE)
    pc = Counts()
    n = lib().uco_simd_sample_run_counts(C.byref(odb.db), C.byref(ix), C.byref(p), threads, q.ctypes.data, len(q), sec,
                                         hits.ctypes.data, cnt.ctypes.data, alns.ctypes.data, C.byref(pc))
    return int(n), sec[0], sec[1], cnt, hits, alns, {f: int(getattr(pc, f)) for f, _ in Counts._fields_}


def simd_align_pairs(odb, p, pairs, threads=0):
    """E5/E6 records of a (query, target) pair list sorted by query, through the SIMD leg (one query group at a time, groups in parallel)"""
    pr = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    out = np.zeros(max(len(pr), 1), ALN_DTYPE)
    L = lib()
    L.uco_simd_align_pairs.argtypes = [C.POINTER(Db), C.c_void_p, C.c_uint64, C.POINTER(Params), C.c_int, C.c_void_p]
    L.uco_simd_align_pairs.restype = None
    L.uco_simd_align_pairs(C.byref(odb.db), pr.ctypes.data, len(pr), C.byref(p), threads, out.ctypes.data)
    return out[: len(pr)]


def simd_cells():
    """(useful, swept) DP cells of the SIMD gapped stage since the last call (resets the counters)"""
    out = (C.c_uint64 * 2)()
    lib().uco_simd_cells(out)
    return int(out[0]), int(out[1])


def align_pair(odb, p, q, t, min_score):
    """the scalar oracle's E5/E6 record for one pair"""
    out = np.zeros(1, ALN_DTYPE)
    lib().uco_align_pair(C.byref(odb.db), int(q), int(t), C.byref(p), int(min_score), out.ctypes.data)
    return out[0]


def min_score(odb, p, q):
    lq = int(odb.db.off[q + 1] - odb.db.off[q])
    return int(lib().uco_min_score(C.byref(p), lq, int(odb.db.off[odb.n])))


def prefilter_shard(odb, p, tbegin=0, tend=None):
    """E1-E4 of every query against the k-mer index of targets [tbegin, tend) -> (counts, hits[n, max_seqs])."""
    ix = build_index(odb, p, tbegin, tend)
    n, M = odb.n, p.max_seqs
    hits = np.zeros((n, M), HIT_DTYPE)
    cnt = np.zeros(n, np.uint32)
    for q in range(n):
        cnt[q] = lib().uco_prefilter_query(C.byref(odb.db), C.byref(ix), q, C.byref(p), hits[q].ctypes.data, None)
    free_index(ix)
    return cnt, hits
