"""unicore_amd — ctypes binding of libunicore_cluster.so (the MI355X-native `unicore cluster` engine).

The library is the product; this module only mirrors its C ABI (include/unicore_cluster.h) for Python
callers (tests, bench.py).  There is no Python or CPU
fallback: if the HIP library cannot be loaded, importing `lib()` raises, and on a machine without a
GPU every compute call returns UC_ERR_DEVICE.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunicore_cluster.so")

UC_OK, UC_ERR_GENERIC, UC_ERR_ARGS, UC_ERR_IO, UC_ERR_DEVICE = 0, 1, 2, 3, 4
NSTAGE = 8
PHASES = ("prefilter", "exchange_lists_to_home", "merge_at_home", "exchange_pairs_to_owner", "install_owned", "gapped", "edge_gather", "rank0_serial_cover")
STAGES = ("load", "index", "kmer", "ungapped", "select", "gapped", "setcover", "output")


class UcOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("threads", C.c_int32), ("verbosity", C.c_int32), ("device", C.c_int32),
                ("num_gpus", C.c_int32), ("cluster_options", C.c_char_p), ("data_dir", C.c_char_p)]


class UcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_seqs", "n_residues", "n_index_entries", "n_sim_kmers", "n_kmer_hits", "n_candidates", "n_prefilter_hits",
        "n_gapped_alignments", "n_start_alignments", "n_pk_reruns", "n_edges", "n_clusters", "cells_fwd", "cells_rev", "cells_start")] + [
        ("algorithmic_bytes", C.c_uint64 * NSTAGE), ("stage_seconds", C.c_double * NSTAGE),
        ("sw_kernel_ms", C.c_double), ("sw_kernel_launches", C.c_uint64), ("sw_algorithmic_bytes", C.c_uint64),
        ("prefilter_kernel_ms", C.c_double), ("n_filtered_hits", C.c_uint64), ("n_sw_runs", C.c_uint64),
        ("cells_run", C.c_uint64), ("exchange_seconds", C.c_double), ("exchange_bytes", C.c_uint64),
        ("n_gpus", C.c_uint32), ("target_shards", C.c_uint32), ("phase_seconds", C.c_double * 8),
        ("nccl_ranks", C.c_uint32), ("reserved0", C.c_uint32), ("exchange2_seconds", C.c_double * 4), ("cells_tb", C.c_uint64)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


class UcT5Stats(C.Structure):
    _fields_ = [("n_seqs", C.c_uint64), ("n_tokens", C.c_uint64), ("flops", C.c_double), ("gpu_ms", C.c_double),
                ("n_replicas", C.c_uint32), ("reserved0", C.c_uint32), ("gpu_ms_sum", C.c_double),
                ("tokens_min_replica", C.c_uint64), ("tokens_max_replica", C.c_uint64)]


HIT_DTYPE = np.dtype([("target", "<u4"), ("score", "<i4"), ("diag", "<i4")])
ALN_DTYPE = np.dtype([(n, "<i4") for n in ("score", "score_rev", "corrected", "qstart", "qend", "tstart", "tend",
                                            "aln_len", "idents", "pass_evalue", "accepted", "gap_opens")])

# every symbol include/unicore_cluster.h declares (tests check the library exports all of them)
SYMBOLS = (
    "uc_cluster", "uc_createtsv", "uc_rmdb", "uc_search", "uc_convertalis", "uc_last_error", "uc_version", "uc_check_options",
    "uc_option_arity", "uc_release_scratch", "uc_createdb", "uc_t5_load", "uc_t5_free", "uc_t5_encode", "uc_t5_get_stats", "uc_comm_unique_id", "uc_comm_create", "uc_comm_destroy", "uc_comm_info", "uc_engine_cluster_step",
    "uc_engine_create", "uc_engine_destroy", "uc_engine_load_db", "uc_engine_set_db", "uc_engine_num_seqs",
    "uc_engine_prefilter", "uc_engine_prefilter_range", "uc_engine_hits_size", "uc_engine_hits_get", "uc_engine_hits_get_range", "uc_engine_hits_set", "uc_engine_hits_merge",
    "uc_engine_hits_export_dev", "uc_engine_hits_import_dev", "uc_engine_setcover",
    "uc_hits_merge", "uc_engine_align", "uc_engine_alns_get", "uc_engine_edges_size", "uc_engine_edges_get",
    "uc_engine_stats", "uc_engine_reset_stats", "uc_setcover", "uc_write_cluster_db",
    "uc_engine_ungapped_batch", "uc_engine_sw_batch", "uc_abi_version", "uc_stats_size", "uc_set_round_hook",
)
ABI_VERSION = 6      # == UC_ABI_VERSION of include/unicore_cluster.h this binding mirrors
ROUND_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_uint32), C.c_int32, C.c_void_p)

_lib = None


class UcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("unicore_cluster error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing — run `make product` (or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    L.uc_last_error.restype = C.c_char_p
    L.uc_version.restype = C.c_char_p
    L.uc_check_options.argtypes = [C.c_char_p]
    L.uc_option_arity.argtypes = [C.c_char_p]
    L.uc_createdb.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_char_p, C.POINTER(UcOpts), C.POINTER(UcT5Stats)]
    L.uc_t5_load.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.uc_t5_free.argtypes = [vp]
    L.uc_t5_free.restype = None
    L.uc_t5_encode.argtypes = [vp, u32, vp, C.c_char_p, vp, vp]
    L.uc_t5_get_stats.argtypes = [vp, C.POINTER(UcT5Stats)]
    L.uc_release_scratch.restype = None
    L.uc_comm_unique_id.argtypes = [vp]
    L.uc_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.uc_comm_destroy.argtypes = [vp]
    L.uc_comm_destroy.restype = None
    L.uc_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.uc_engine_cluster_step.argtypes = [vp, vp, i32, vp, C.POINTER(u64)]
    L.uc_cluster.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(UcOpts), C.POINTER(UcStats)]
    L.uc_createtsv.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(UcOpts)]
    L.uc_search.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(UcOpts), C.POINTER(UcStats)]
    L.uc_convertalis.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(UcOpts)]
    L.uc_engine_prefilter_range.argtypes = [vp, u32, u32, u32, u32]
    L.uc_rmdb.argtypes = [C.c_char_p]
    L.uc_engine_create.argtypes = [C.POINTER(UcOpts), C.POINTER(vp)]
    L.uc_engine_destroy.argtypes = [vp]
    L.uc_engine_destroy.restype = None
    L.uc_engine_load_db.argtypes = [vp, C.c_char_p]
    L.uc_engine_set_db.argtypes = [vp, u32, vp, vp, vp]
    L.uc_engine_num_seqs.argtypes = [vp]
    L.uc_engine_num_seqs.restype = u32
    L.uc_engine_prefilter.argtypes = [vp, u32, u32]
    L.uc_engine_hits_size.argtypes = [vp, C.POINTER(u64)]
    L.uc_engine_hits_get.argtypes = [vp, vp, vp]
    L.uc_engine_hits_get_range.argtypes = [vp, u32, u32, vp, vp]
    L.uc_engine_hits_set.argtypes = [vp, vp, vp]
    L.uc_engine_hits_merge.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]
    L.uc_engine_hits_export_dev.argtypes = [vp, vp, vp, vp, vp]
    L.uc_engine_setcover.argtypes = [vp, vp, u64, vp]
    L.uc_engine_hits_import_dev.argtypes = [vp, u64, vp, vp, vp, vp, u32, u32, C.POINTER(u64)]
    L.uc_hits_merge.argtypes = [u32, i32, C.c_int, C.POINTER(vp), C.POINTER(vp), vp, vp, u64, C.POINTER(u64)]
    L.uc_engine_align.argtypes = [vp, u32, u32]
    L.uc_engine_alns_get.argtypes = [vp, u32, u32, vp]
    L.uc_engine_edges_size.argtypes = [vp, C.POINTER(u64)]
    L.uc_engine_edges_get.argtypes = [vp, vp]
    L.uc_engine_stats.argtypes = [vp, C.POINTER(UcStats)]
    L.uc_engine_reset_stats.argtypes = [vp]
    L.uc_engine_reset_stats.restype = None
    L.uc_setcover.argtypes = [u32, vp, u64, vp]
    L.uc_write_cluster_db.argtypes = [C.c_char_p, u32, vp]
    L.uc_engine_ungapped_batch.argtypes = [vp, u64, vp, vp, vp, vp]
    L.uc_engine_sw_batch.argtypes = [vp, C.c_int, u64, vp, vp, vp, vp, vp, vp, vp]
    L.uc_abi_version.restype = u32
    L.uc_stats_size.restype = C.c_size_t
    L.uc_set_round_hook.argtypes = [ROUND_HOOK, vp]
    L.uc_set_round_hook.restype = None
    # uc_stats carries no size field: a binding that disagrees with the library about its layout would be overrun silently
    if L.uc_abi_version() != ABI_VERSION or L.uc_stats_size() != C.sizeof(UcStats):
        raise ImportError("libunicore_cluster.so has ABI %d / uc_stats of %d bytes; this binding mirrors ABI %d / %d bytes — rebuild (`make product`)"
                          % (L.uc_abi_version(), L.uc_stats_size(), ABI_VERSION, C.sizeof(UcStats)))
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise UcError(rc, lib().uc_last_error().decode(errors="replace"))


def make_opts(cluster_options="", threads=1, verbosity=1, device=-1, data_dir=None, num_gpus=1):
    o = UcOpts()
    o.struct_size = C.sizeof(UcOpts)
    o.threads = threads
    o.verbosity = verbosity
    o.device = device
    o.num_gpus = num_gpus
    o.cluster_options = cluster_options.encode()
    o.data_dir = data_dir.encode() if data_dir else None
    return o


def version():
    return lib().uc_version().decode()


def check_options(s):
    return lib().uc_check_options(s.encode())


def cluster(db, out_cluster_db, tmp, cluster_options="-c 0.8", threads=1, verbosity=1, device=-1, num_gpus=1):
    """== `foldseek cluster` (cluster.rs:45-56).  Returns the stats dict.  num_gpus: 0 = all visible GPUs."""
    o, st = make_opts(cluster_options, threads, verbosity, device, num_gpus=num_gpus), UcStats()
    _check(lib().uc_cluster(db.encode(), out_cluster_db.encode(), tmp.encode(), C.byref(o), C.byref(st)))
    return st.as_dict()


def createtsv(db, cluster_db, out_tsv, verbosity=1):
    o = make_opts("", 1, verbosity)
    _check(lib().uc_createtsv(db.encode(), cluster_db.encode(), out_tsv.encode(), C.byref(o)))


def search(query_db, target_db, out_aln_db, tmp, search_options="-c 0.8", threads=1, verbosity=1, device=-1):
    """== `foldseek search` (search.rs:44-50).  Returns the stats dict."""
    o, st = make_opts(search_options, threads, verbosity, device), UcStats()
    _check(lib().uc_search(query_db.encode(), target_db.encode(), out_aln_db.encode(), tmp.encode(), C.byref(o), C.byref(st)))
    return st.as_dict()


def convertalis(query_db, target_db, aln_db, out_m8, verbosity=1):
    """== `foldseek convertalis` (search.rs:57-60)"""
    o = make_opts("", 1, verbosity)
    _check(lib().uc_convertalis(query_db.encode(), target_db.encode(), aln_db.encode(), out_m8.encode(), C.byref(o)))


def rmdb(prefix):
    _check(lib().uc_rmdb(prefix.encode()))


def setcover(n, edges):
    e = np.ascontiguousarray(edges, np.uint32).reshape(-1, 2)
    assign = np.zeros(n, np.uint32)
    _check(lib().uc_setcover(n, e.ctypes.data, len(e), assign.ctypes.data))
    return assign


def hits_merge(n_seqs, max_seqs, parts):
    """parts: list of (counts u32[n_seqs], hits HIT_DTYPE[...]).  Host-only (no device needed)."""
    k = len(parts)
    cs = [np.ascontiguousarray(c, np.uint32) for c, _ in parts]
    hs = [np.ascontiguousarray(h, HIT_DTYPE) for _, h in parts]
    cp = (C.c_void_p * k)(*[c.ctypes.data for c in cs])
    hp = (C.c_void_p * k)(*[h.ctypes.data for h in hs])
    cap = int(sum(len(h) for h in hs))
    oc, oh, on = np.zeros(n_seqs, np.uint32), np.zeros(max(cap, 1), HIT_DTYPE), C.c_uint64()
    _check(lib().uc_hits_merge(n_seqs, max_seqs, k, cp, hp, oc.ctypes.data, oh.ctypes.data, cap, C.byref(on)))
    return oc, oh[: on.value].copy()


def createdb(fasta_paths, out_db, model, verbosity=1, device=-1, num_gpus=1):
    """== `foldseek createdb <fasta...> <db> --prostt5-model <model>` (createdb.rs:157-166): ProstT5 AA -> 3Di on the GPU(s);
    num_gpus: one encoder replica per GPU, sequences sharded over them (0 = all visible GPUs)"""
    if isinstance(fasta_paths, str):
        fasta_paths = [fasta_paths]
    arr = (C.c_char_p * len(fasta_paths))(*[p.encode() for p in fasta_paths])
    o, st = make_opts("", 1, verbosity, device, None, num_gpus), UcT5Stats()
    _check(lib().uc_createdb(arr, len(fasta_paths), out_db.encode(), model.encode(), C.byref(o), C.byref(st)))
    return {k: getattr(st, k) for k, _ in UcT5Stats._fields_}


_round_hook_keepalive = None


def set_round_hook(fn):
    """uc_set_round_hook: fn(round, ids, kmer_thr, view) is called once per workflow round of uc_cluster — after the round's gapped stage,
    before its set cover — with round = -1 for the linear-time pre-step, ids = database sequence number of every round-local index (a copy)
    and view = an Engine facade over the round's engine (hits_range / alns_range / stats; valid only during the call).  None unregisters."""
    global _round_hook_keepalive
    if fn is None:
        lib().uc_set_round_hook(ROUND_HOOK(), None)
        _round_hook_keepalive = None
        return

    def tramp(_user, rnd, m, ids, kthr, eng):
        view = Engine.__new__(Engine)
        view._h = C.c_void_p(eng)
        try:
            fn(int(rnd), np.ctypeslib.as_array(ids, shape=(int(m),)).copy(), int(kthr), view)
        finally:
            view._h = C.c_void_p()      # a view never destroys the engine it looked at
    cb = ROUND_HOOK(tramp)
    _round_hook_keepalive = cb
    lib().uc_set_round_hook(cb, None)


class T5Encoder:
    """the ProstT5 AA -> 3Di encoder (uc_t5_* of the C ABI)"""

    def __init__(self, model, device=-1):
        self._h = C.c_void_p()
        _check(lib().uc_t5_load(model.encode(), device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().uc_t5_free(self._h)
            self._h = C.c_void_p()

    def encode(self, seqs, logits=False):
        """seqs: list of residue strings -> list of uint8 arrays (3Di states 0..19) [, list of float32 [L, 20] logits]"""
        off = np.zeros(len(seqs) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        aa = "".join(seqs).encode()
        codes = np.zeros(max(int(off[-1]), 1), np.uint8)
        lg = np.zeros((max(int(off[-1]), 1), 20), np.float32) if logits else None
        _check(lib().uc_t5_encode(self._h, len(seqs), off.ctypes.data, aa, codes.ctypes.data, lg.ctypes.data if logits else None))
        out = [codes[int(off[i]):int(off[i + 1])].copy() for i in range(len(seqs))]
        if logits:
            return out, [lg[int(off[i]):int(off[i + 1])].copy() for i in range(len(seqs))]
        return out

    def stats(self):
        st = UcT5Stats()
        _check(lib().uc_t5_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in UcT5Stats._fields_}


class Comm:
    """RCCL communicator of the one-process-per-GPU layout (uc_comm_* of the C ABI).  Rank 0 calls Comm.unique_id()
    and ships the 128 bytes to the other ranks; creating the communicator is collective."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _check(lib().uc_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid, rank, world, device=-1):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        _check(lib().uc_comm_create(buf, rank, world, device, C.byref(self._h)))

    def info(self):
        """(ranks, rank, device) as RCCL itself reports them (ncclCommCount / ncclCommUserRank / ncclCommCuDevice)"""
        n, r, d = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().uc_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d)))
        return n.value, r.value, d.value

    def close(self):
        if self._h:
            lib().uc_comm_destroy(self._h)
            self._h = C.c_void_p()


class Engine:
    """One engine = one HIP device with the sequence DB resident in HBM (uc_engine_* of the C ABI)."""

    def __init__(self, cluster_options="-c 0.8", threads=1, verbosity=1, device=-1, data_dir=None):
        self._h = C.c_void_p()
        o = make_opts(cluster_options, threads, verbosity, device, data_dir)
        _check(lib().uc_engine_create(C.byref(o), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().uc_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_db(self, prefix):
        _check(lib().uc_engine_load_db(self._h, prefix.encode()))

    def set_db(self, off, s3, sa):
        off = np.ascontiguousarray(off, np.uint64)
        s3 = np.ascontiguousarray(s3, np.uint8)
        sa = np.ascontiguousarray(sa, np.uint8)
        _check(lib().uc_engine_set_db(self._h, len(off) - 1, off.ctypes.data, s3.ctypes.data, sa.ctypes.data))

    @property
    def n(self):
        return int(lib().uc_engine_num_seqs(self._h))

    def prefilter(self, tbegin=0, tend=None, qbegin=None, qend=None):
        tend = self.n if tend is None else tend
        if qbegin is None and qend is None:
            _check(lib().uc_engine_prefilter(self._h, tbegin, tend))
        else:
            _check(lib().uc_engine_prefilter_range(self._h, tbegin, tend, qbegin or 0, self.n if qend is None else qend))

    def hits_size(self):
        nh = C.c_uint64()
        _check(lib().uc_engine_hits_size(self._h, C.byref(nh)))
        return int(nh.value)

    def hits(self):
        nh = C.c_uint64()
        _check(lib().uc_engine_hits_size(self._h, C.byref(nh)))
        counts = np.zeros(self.n, np.uint32)
        hits = np.zeros(max(nh.value, 1), HIT_DTYPE)
        _check(lib().uc_engine_hits_get(self._h, counts.ctypes.data, hits.ctypes.data))
        return counts, hits[: nh.value]

    def hits_range(self, qbegin, qend):
        """(counts, hits) of the queries [qbegin, qend) only"""
        counts = np.zeros(max(qend - qbegin, 1), np.uint32)
        _check(lib().uc_engine_hits_get_range(self._h, qbegin, qend, counts.ctypes.data, None))
        counts = counts[: qend - qbegin]
        k = int(counts.sum())
        hits = np.zeros(max(k, 1), HIT_DTYPE)
        _check(lib().uc_engine_hits_get_range(self._h, qbegin, qend, None, hits.ctypes.data))
        return counts, hits[:k]

    def alns_range(self, qbegin, qend):
        """alignment records of the queries [qbegin, qend) only (one per hit, in hit order)"""
        counts, _ = np.zeros(max(qend - qbegin, 1), np.uint32), None
        _check(lib().uc_engine_hits_get_range(self._h, qbegin, qend, counts.ctypes.data, None))
        k = int(counts[: qend - qbegin].sum())
        out = np.zeros(max(k, 1), ALN_DTYPE)
        _check(lib().uc_engine_alns_get(self._h, qbegin, qend, out.ctypes.data))
        return out[:k]

    def set_hits(self, counts, hits):
        counts = np.ascontiguousarray(counts, np.uint32)
        hits = np.ascontiguousarray(hits, HIT_DTYPE)
        _check(lib().uc_engine_hits_set(self._h, counts.ctypes.data, hits.ctypes.data))

    def hits_export_dev(self, d_query, d_target, d_score, d_diag):
        """copy the device-resident hit lists into caller-owned DEVICE buffers (raw pointers, hits_size() x 4 B each)"""
        _check(lib().uc_engine_hits_export_dev(self._h, d_query, d_target, d_score, d_diag))

    def hits_import_dev(self, n, d_query, d_target, d_score, d_diag, rank=0, world=1):
        """install the merged union of shard lists given as DEVICE arrays; keeps the pairs owned by `rank`"""
        k = C.c_uint64()
        _check(lib().uc_engine_hits_import_dev(self._h, n, d_query, d_target, d_score, d_diag, rank, world, C.byref(k)))
        return int(k.value)

    def setcover(self, edges):
        """E7 with the graph built on this engine's GPU and the greedy cover on the host (same result as setcover())"""
        e = np.ascontiguousarray(edges, np.uint32).reshape(-1, 2)
        assign = np.zeros(self.n, np.uint32)
        _check(lib().uc_engine_setcover(self._h, e.ctypes.data, len(e), assign.ctypes.data))
        return assign

    def align(self, qbegin=0, qend=None):
        _check(lib().uc_engine_align(self._h, qbegin, self.n if qend is None else qend))

    def cluster_step(self, comm=None, target_shards=0):
        """One pass of the (sharded) hot path inside the library: prefilter of this rank's grid cell -> RCCL hit all-gather +
        device merge -> E5/E6 -> edges to rank 0 -> set cover there.  Returns (assign or None, gapped alignments of this rank)."""
        rank0 = comm is None or comm.rank == 0
        assign = np.zeros(self.n, np.uint32) if rank0 else None
        k = C.c_uint64()
        _check(lib().uc_engine_cluster_step(self._h, comm._h if comm is not None else None, target_shards,
                                            assign.ctypes.data if rank0 else None, C.byref(k)))
        return assign, int(k.value)

    def alns(self, qbegin=0, qend=None):
        qend = self.n if qend is None else qend
        counts, _ = self.hits()
        n = int(counts[qbegin:qend].sum())
        out = np.zeros(max(n, 1), ALN_DTYPE)
        _check(lib().uc_engine_alns_get(self._h, qbegin, qend, out.ctypes.data))
        return out[:n]

    def edges(self):
        ne = C.c_uint64()
        _check(lib().uc_engine_edges_size(self._h, C.byref(ne)))
        e = np.zeros((max(ne.value, 1), 2), np.uint32)
        _check(lib().uc_engine_edges_get(self._h, e.ctypes.data))
        return e[: ne.value]

    def stats(self):
        st = UcStats()
        _check(lib().uc_engine_stats(self._h, C.byref(st)))
        return st.as_dict()

    def reset_stats(self):
        lib().uc_engine_reset_stats(self._h)

    def ungapped(self, q, t, diag):
        q = np.ascontiguousarray(q, np.uint32); t = np.ascontiguousarray(t, np.uint32)
        diag = np.ascontiguousarray(diag, np.int32)
        out = np.zeros(len(q), np.int32)
        _check(lib().uc_engine_ungapped_batch(self._h, len(q), q.ctypes.data, t.ctypes.data, diag.ctypes.data, out.ctypes.data))
        return out

    def sw(self, mode, q, t, qend=None, tend=None):
        q = np.ascontiguousarray(q, np.uint32); t = np.ascontiguousarray(t, np.uint32)
        n = len(q)
        qe = np.ascontiguousarray(qend, np.int32) if qend is not None else None
        te = np.ascontiguousarray(tend, np.int32) if tend is not None else None
        s, oq, ot = np.zeros(n, np.int32), np.full(n, -1, np.int32), np.full(n, -1, np.int32)
        _check(lib().uc_engine_sw_batch(self._h, mode, n, q.ctypes.data, t.ctypes.data,
                                        qe.ctypes.data if qe is not None else None, te.ctypes.data if te is not None else None,
                                        s.ctypes.data, oq.ctypes.data, ot.ctypes.data))
        return s, oq, ot
