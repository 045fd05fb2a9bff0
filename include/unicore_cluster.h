/* unicore_cluster.h — C ABI of the MI355X-native `unicore cluster` engine (libunicore_cluster.so).
 *
 * This is the drop-in boundary for the ONE hot path of steineggerlab/unicore: the three external
 * `foldseek` invocations of /root/reference/src/modules/cluster.rs:45-76.  The reference has no
 * in-process FFI for this path (its extension point is the binary registry path.cfg:2 +
 * src/util/command.rs:4-24), so the entry points below are exactly what a Rust `unicore` would bind
 * with `extern "C"` to replace those three spawns (INTEGRATION.md shows the binding), plus a staged
 * engine API used by the multi-GPU driver, bench.py and the parity tests.
 *
 * Conventions: plain pointers and sizes, caller-owned memory unless stated, UTF-8 paths, no
 * exceptions cross the boundary.  Every int-returning function returns 0 on success or an error class
 * (exit status is the reference's only error channel: src/util/command.rs:10-14):
 *     1 generic, 2 bad arguments / unknown flag, 3 I/O, 4 device (HIP) failure or no GPU.
 * uc_last_error() gives the message (thread-local).  There is NO CPU fallback: without a working HIP
 * device every compute entry point fails with class 4.
 */
#ifndef UNICORE_CLUSTER_H
#define UNICORE_CLUSTER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UC_OK 0
#define UC_ERR_GENERIC 1
#define UC_ERR_ARGS 2
#define UC_ERR_IO 3
#define UC_ERR_DEVICE 4

/* Options.  `cluster_options` is the raw whitespace-split Foldseek-style flag string that
 * src/modules/cluster.rs:35,49 forwards verbatim (default "-c 0.8", src/util/arg_parser.rs:238-239).
 * Unknown flags are rejected (UC_ERR_ARGS), never ignored. */
typedef struct uc_opts {
    uint32_t struct_size;        /* = sizeof(uc_opts) */
    int32_t threads;             /* host threads, >=1 (cluster.rs:46 "--threads") */
    int32_t verbosity;           /* 0..3, Foldseek scale after the 4->3,3->2 mapping of cluster.rs:18 */
    int32_t device;              /* HIP device ordinal of a single-GPU run / of an engine, -1 = current */
    int32_t num_gpus;            /* uc_cluster, uc_createdb: GPUs the run is spread over: 0 = all visible, 1 = `device`,
                                    N = devices 0..N-1; "--gpus N" inside cluster_options overrides it (uc_cluster) */
    const char *cluster_options; /* may be NULL == "" */
    const char *data_dir;        /* directory holding mat3di.out / blosum62.out; NULL = <lib dir>/data.  Without a real
                                    mat3di.out the call fails unless UC_ALLOW_SYNTHETIC=1 opts into the seeded stand-in */
} uc_opts;

/* ABI revision of this header: bumped whenever a struct below grows or an entry point changes meaning.  uc_stats is written in full by
 * uc_cluster / uc_search / uc_engine_stats and carries no size field of its own, so a caller built against an older header must check
 * uc_abi_version() == UC_ABI_VERSION (or uc_stats_size() == sizeof(uc_stats)) before passing one in. */
#define UC_ABI_VERSION 6
uint32_t uc_abi_version(void);
size_t uc_stats_size(void);

#define UC_NSTAGE 8
enum { UC_ST_LOAD = 0, UC_ST_INDEX = 1, UC_ST_KMER = 2, UC_ST_UNGAPPED = 3, UC_ST_SELECT = 4,
       UC_ST_GAPPED = 5, UC_ST_SETCOVER = 6, UC_ST_OUTPUT = 7 };

#define UC_NPHASE 8
typedef struct uc_stats {
    uint64_t n_seqs, n_residues;
    uint64_t n_index_entries, n_sim_kmers, n_kmer_hits, n_candidates, n_prefilter_hits;
    uint64_t n_gapped_alignments;        /* (query,target) pairs handed to stage E5: the metric's unit */
    uint64_t n_start_alignments;         /* pairs that also ran the start-position pass */
    uint64_t n_pk_reruns;                /* alignment-passes of the packed 16-bit kernel re-run in int32 */
    uint64_t n_edges, n_clusters;
    uint64_t cells_fwd, cells_rev, cells_start;          /* DP cell updates per pass */
    uint64_t algorithmic_bytes[UC_NSTAGE];               /* SURVEY.md 8(d) per-stage algorithmic bytes */
    double stage_seconds[UC_NSTAGE];                     /* host wall per stage */
    /* dominant kernel (gapped SW, all passes): HIP-event time on the engine stream */
    double sw_kernel_ms;
    uint64_t sw_kernel_launches;
    uint64_t sw_algorithmic_bytes;                       /* sum over launches of 2*(Lq+Lt)+32 per alignment-pass */
    double prefilter_kernel_ms;                          /* all prefilter kernels (HIP events) */
    uint64_t n_filtered_hits;                            /* k-mer hits that survive the double-hit filter and get sorted */
    uint64_t n_sw_runs;                                  /* DP problems actually executed over all passes (mutual hits share one, re-runs add) */
    uint64_t cells_run;                                  /* DP cell updates actually executed (cells_* above are the algorithmic counts) */
    /* multi-GPU exchange: wall seconds of the two hit-list exchanges (shard lists to the query's home rank, surviving
     * pairs to their owner rank) + device merges + edge gather on this rank, and the bytes this rank received from its peers */
    double exchange_seconds;
    uint64_t exchange_bytes;
    uint32_t n_gpus, target_shards;                      /* the Q x T grid the run used: Q = n_gpus / target_shards */
    /* N > 1 only: wall seconds of this rank per phase of the sharded pass (uc_cluster reports the slowest rank per phase):
     * [0] prefilter of the rank's grid cell, [1] exchange 1 (shard lists to the query's home rank), [2] merge + top-M at home,
     * [3] partition by owner + exchange 2 (surviving pairs to their owner rank), [4] install of the owned lists, [5] gapped stage,
     * [6] edge gather to rank 0, [7] rank 0's serial tail (graph + greedy cover) */
    double phase_seconds[UC_NPHASE];
    uint32_t nccl_ranks, reserved0;                      /* ncclCommCount of the run's communicator (0 = no RCCL: one GPU or virtual ranks) */
    /* phase [3] taken apart (ABI 5): [0] stable partition of the merged pairs by owner rank (device sort), [1] exchange of the per-peer counts
     * (a small all-gather: it ends when the SLOWEST rank has finished its partition), [2] waiting at the rendezvous of the data exchange
     * (in-process ranks: the barriers around the device copies; RCCL ranks: 0 - the wait is inside the stream synchronisation of [3]),
     * [3] the data movement itself (grouped ncclSend / ncclRecv + stream synchronisation, or the device copies of in-process ranks) */
    double exchange2_seconds[4];
    /* (ABI 5) DP cells of the traceback boxes [qStart..qEnd] x [tStart..tEnd] of every pair whose statistics were asked for (--min-seq-id, search):
     * the fourth pass of the spec; cells_fwd + cells_rev + cells_start + cells_tb is what cells_run has to be read against */
    uint64_t cells_tb;
} uc_stats;

/* ---- the three calls of cluster.rs ------------------------------------------------------------ */
/* == `foldseek cluster --threads T -v V <db> <out>_cluster <tmp> <opts...>`  (cluster.rs:45-56) */
int uc_cluster(const char *db, const char *out_cluster_db, const char *tmp, const uc_opts *o, uc_stats *stats_out);
/* == `foldseek createtsv --threads T -v V <db> <db> <out>_cluster <out>.tsv`   (cluster.rs:59-64) */
int uc_createtsv(const char *db, const char *cluster_db, const char *out_tsv, const uc_opts *o);
/* == `foldseek rmdb <out>_cluster -v V`                                         (cluster.rs:67-76) */
int uc_rmdb(const char *db_prefix);

/* Workflow observer (optional).  A bare "-c 0.8" — what cluster.rs:35,49 forwards — runs the DEFAULT workflow: a linear-time pre-step and a
 * 3-step cascade, each round on the representatives of the one before.  A registered hook is called by uc_cluster once per round, after the
 * round's gapped stage and before its set cover, on the calling thread (single-GPU runs; with N GPUs: rank 0's thread, and `round_engine`
 * holds rank 0's share of the pairs only).  round = -1 for the pre-step, 0.. for the cascade rounds; the round works on n_round_seqs
 * sequences, round-local index i = database sequence seq_ids[i]; kmer_thr is the round's k-mer score threshold (the sensitivity rises from
 * 1 to the target over the rounds).  round_engine is valid only during the call: its hit lists and alignment records are the round's
 * (uc_engine_hits_get_range / uc_engine_alns_get / uc_engine_stats, round-local indices).  The at-size parity tests sample every round
 * against the CPU oracle through this; bench.py uses it to size the CPU baseline of the workflow round by round.  NULL unregisters. */
typedef struct uc_engine uc_engine;
typedef void (*uc_round_hook)(void *user, int32_t round, uint32_t n_round_seqs, const uint32_t *seq_ids, int32_t kmer_thr, uc_engine *round_engine);
void uc_set_round_hook(uc_round_hook hook, void *user);

/* ---- the calls of src/modules/search.rs, same kernels, query DB vs target DB --------------- */
/* == `foldseek search --threads T <queryDB> <targetDB> <out>_aln <tmp> <opts...>`  (search.rs:44-50; note that Unicore
 *    passes its TARGET argument first, i.e. as Foldseek's query DB).  Defaults as for cluster except -e 10 and
 *    --max-seqs 1000; traceback statistics for every accepted pair. */
int uc_search(const char *query_db, const char *target_db, const char *out_aln_db, const char *tmp, const uc_opts *o, uc_stats *stats_out);
/* == `foldseek convertalis --threads T <queryDB> <targetDB> <out>_aln <out>.m8`    (search.rs:52-57) */
int uc_convertalis(const char *query_db, const char *target_db, const char *aln_db, const char *out_m8, const uc_opts *o);

/* ---- createdb's GPU stage (BASELINE configs[4]): ProstT5 AA -> 3Di on the matrix cores ---------------
 * == `foldseek createdb <fasta> <db> --prostt5-model <dir> [--gpu 1]` (src/modules/createdb.rs:157-166).  `model` is the
 * GGUF file or the directory holding prostt5-f16.gguf (createdb.rs:148).  Writes <db>, <db>_h, <db>_ss (predicted 3Di) with
 * their .index / .dbtype and <db>.lookup: the files uc_cluster / uc_search read.  Several FASTA files: one path per entry. */
typedef struct uc_t5_stats {
    uint64_t n_seqs, n_tokens;       /* tokens = residues + 2 per sequence (<AA2fold> ... </s>) */
    double flops;                    /* algorithmic FLOPs of the linear layers + attention */
    double gpu_ms;                   /* HIP-event time of the encoder passes (several replicas: the SLOWEST replica's - they run side by side) */
    /* (ABI 6) uc_createdb on N GPUs: one encoder replica per GPU, sequences sharded over them, no collective (uc_opts.num_gpus as for uc_cluster) */
    uint32_t n_replicas, reserved0;
    double gpu_ms_sum;               /* sum over the replicas (== gpu_ms for one) */
    uint64_t tokens_min_replica, tokens_max_replica;   /* balance of the dynamic dealing */
} uc_t5_stats;
int uc_createdb(const char *const *fasta_paths, int n_fasta, const char *out_db, const char *model, const uc_opts *o, uc_t5_stats *stats_out);
/* the encoder alone (no disk round trip: the codes go straight into uc_engine_set_db): */
typedef struct uc_t5 uc_t5;
int uc_t5_load(const char *model, int32_t device, uc_t5 **out);
void uc_t5_free(uc_t5 *m);
/* n sequences as residue letters, off[n + 1] byte offsets into aa; codes (one 3Di state 0..19 per residue, same offsets);
 * logits (nullable): 20 floats per residue */
int uc_t5_encode(uc_t5 *m, uint32_t n, const uint64_t *off, const char *aa, uint8_t *codes, float *logits);
int uc_t5_get_stats(const uc_t5 *m, uc_t5_stats *out);

const char *uc_last_error(void);
const char *uc_version(void);
/* validates a Foldseek-style option string without running anything (0 or UC_ERR_ARGS) */
int uc_check_options(const char *cluster_options);
/* how the engine's flag table reads one flag: 1 = takes a value, 2 = switch with an optional 0/1, -1 = unknown.  The argv
 * shim uses it to split Foldseek-style command lines with flags anywhere. */
int uc_option_arity(const char *flag);
/* Work buffers (tens of GB of HBM at bench scale) are parked per device when an engine is destroyed, so that the next
 * uc_cluster / engine of the process does not pay hipMalloc again (UC_KEEP_SCRATCH=0 disables parking).  An in-process
 * host that is done with the engine for now calls this to give the memory back.  Loading the library also sets
 * GPU_MAX_HW_QUEUES=8 in the process environment if the variable is unset (the class kernels of a pass run on 8 streams). */
void uc_release_scratch(void);

/* ---- staged engine API (multi-GPU driver, bench, parity tests) -------------------------------- */

typedef struct uc_hit {          /* one prefilter result (stage E4 output) */
    uint32_t target;
    int32_t score;               /* ungapped diagonal score, 0..255 */
    int32_t diag;                /* query pos - target pos of the best diagonal */
} uc_hit;

typedef struct uc_aln {          /* one gapped alignment result (stage E5/E6 output) */
    int32_t score, score_rev, corrected;
    int32_t qstart, qend, tstart, tend;    /* valid iff pass_evalue */
    int32_t aln_len, idents;               /* valid iff a seq-id threshold is active */
    int32_t pass_evalue, accepted;
    int32_t gap_opens;                     /* with aln_len/idents: number of gaps on the traceback (search path) */
} uc_aln;

int uc_engine_create(const uc_opts *o, uc_engine **out);
void uc_engine_destroy(uc_engine *e);
/* load <db>, <db>_ss, <db>_h (+ .index) from disk and upload both tracks to HBM */
int uc_engine_load_db(uc_engine *e, const char *db_prefix);
/* or hand over encoded sequences directly: codes 0..20, off has n+1 entries (bytes), no padding */
int uc_engine_set_db(uc_engine *e, uint32_t n, const uint64_t *off, const uint8_t *s3, const uint8_t *sa);
uint32_t uc_engine_num_seqs(const uc_engine *e);

/* E1-E4: index targets [tbegin,tend), match ALL queries against it, keep per-query top max_seqs */
int uc_engine_prefilter(uc_engine *e, uint32_t tbegin, uint32_t tend);
/* the same with the queries restricted to [qbegin,qend): query DB vs target DB inside one loaded set (search path) */
int uc_engine_prefilter_range(uc_engine *e, uint32_t tbegin, uint32_t tend, uint32_t qbegin, uint32_t qend);
/* hit lists live in the engine: counts[n_seqs], hits flat, grouped by query in query order */
int uc_engine_hits_size(const uc_engine *e, uint64_t *n_hits);
int uc_engine_hits_get(const uc_engine *e, uint32_t *counts, uc_hit *hits);
/* the same for the queries [qbegin,qend) only: counts may be NULL; hits must hold the sum of their counts (at BASELINE
 * configs[2] scale the whole list is gigabytes, a sample of queries is not) */
int uc_engine_hits_get_range(const uc_engine *e, uint32_t qbegin, uint32_t qend, uint32_t *counts, uc_hit *hits);
/* replace the engine's hit lists (counts[n_seqs] + flat hits grouped by query) */
int uc_engine_hits_set(uc_engine *e, const uint32_t *counts, const uc_hit *hits);
/* replace the engine's hit lists by the merge of n_parts shard lists (each: counts[n_seqs] + flat hits),
 * keeping per query the top max_seqs under the frozen order (score desc, target asc) */
int uc_engine_hits_merge(uc_engine *e, int n_parts, const uint32_t *const *counts, const uc_hit *const *hits);
/* the same merge as a free host function (needs no device): the post-exchange step of the multi-GPU
 * layout.  out_hits must hold out_capacity entries; *out_n receives the merged total. */
int uc_hits_merge(uint32_t n_seqs, int32_t max_seqs, int n_parts, const uint32_t *const *counts,
                  const uc_hit *const *hits, uint32_t *out_counts, uc_hit *out_hits, uint64_t out_capacity,
                  uint64_t *out_n);

/* ---- device-resident exchange for the multi-GPU layout: the hit lists never visit the host.
 * export: copy the engine's hit lists (hits_size elements per array, grouped by query) into caller-provided
 *         DEVICE buffers (e.g. the storage of a torch tensor that RCCL then all-gathers).
 * import: install the union of all shards' lists given as DEVICE arrays in any order: merge per query under the
 *         frozen order (score desc, target asc), keep max_seqs, and - if world > 1 - keep only the pairs owned by
 *         `rank`.  Ownership is a hash of the UNORDERED pair, so (q,t) and (t,q) land on the same rank and share
 *         their DP there; over all ranks every merged pair is aligned exactly once.  *n_kept = pairs installed. */
int uc_engine_hits_export_dev(const uc_engine *e, uint32_t *d_query, uint32_t *d_target, int32_t *d_score, int32_t *d_diag);
int uc_engine_hits_import_dev(uc_engine *e, uint64_t n, const uint32_t *d_query, const uint32_t *d_target, const int32_t *d_score,
                              const int32_t *d_diag, uint32_t rank, uint32_t world, uint64_t *n_kept);

/* ---- one process per GPU: the same sharded pass driven from outside (bench.py under torch.distributed.run) ----------
 * The data path stays inside the library: two ragged RCCL exchanges of the hit lists (grouped ncclSend / ncclRecv), device merges,
 * point-to-point edge gather.
 * Rank 0 creates an id and ships its 128 bytes to the other ranks by any means (a file, torch.distributed's store);
 * uc_comm_create is collective (ncclCommInitRank) and binds the communicator to HIP device `device`. */
typedef struct uc_comm uc_comm;
#define UC_COMM_ID_BYTES 128
int uc_comm_unique_id(uint8_t id[UC_COMM_ID_BYTES]);
int uc_comm_create(const uint8_t id[UC_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, uc_comm **out);
void uc_comm_destroy(uc_comm *c);
/* what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): the number of ranks
 * it was built over, this rank, its HIP device.  bench.py prints the count in its line; the N > 1 tests assert it == N. */
int uc_comm_info(const uc_comm *c, int32_t *nccl_ranks, int32_t *nccl_rank, int32_t *device);
/* One pass of the hot path on this rank: E1-E4 on the rank's cell of the Q x T grid (target_shards = T, 0 = one target
 * shard per GPU: the north-star layout) -> exchange 1 + merge at the home rank -> exchange 2 to the owner rank -> E5/E6 on the rank's pairs -> edges
 * to rank 0 -> set cover on rank 0.  comm == NULL runs the single-GPU pass.  assign[n_seqs] is written on rank 0 only
 * (may be NULL elsewhere); *n_alignments = gapped alignments of THIS rank. */
int uc_engine_cluster_step(uc_engine *e, uc_comm *comm, int32_t target_shards, uint32_t *assign, uint64_t *n_alignments);

/* E5-E6 for queries [qbegin,qend) of the engine's hit lists; results are kept per hit */
int uc_engine_align(uc_engine *e, uint32_t qbegin, uint32_t qend);
int uc_engine_alns_get(const uc_engine *e, uint32_t qbegin, uint32_t qend, uc_aln *out);   /* one per hit */
int uc_engine_edges_size(const uc_engine *e, uint64_t *n_edges);
int uc_engine_edges_get(const uc_engine *e, uint32_t *edges /* 2*n_edges: (query,target) */);

int uc_engine_stats(const uc_engine *e, uc_stats *out);
void uc_engine_reset_stats(uc_engine *e);

/* E7 (host, as the north star prescribes): greedy set cover over (a,b) pairs; assign[i] = representative */
int uc_setcover(uint32_t n, const uint32_t *edges, uint64_t n_edges, uint32_t *assign);
/* the same result with the graph built on the engine's GPU (sort + unique of the edge list) and only the greedy
 * cover on the host: what uc_cluster and the bench step use */
int uc_engine_setcover(uc_engine *e, const uint32_t *edges, uint64_t n_edges, uint32_t *assign);
/* E8/E9 outputs from an assignment: cluster DB (<prefix>, .index, .dbtype) */
int uc_write_cluster_db(const char *out_cluster_db, uint32_t n, const uint32_t *assign);

/* ---- kernel-level entry points (parity tests call the HIP kernels through these) -------------- */
/* ungapped diagonal score (E3) for n candidates (q[i], t[i], diag[i]) of the engine's DB */
int uc_engine_ungapped_batch(uc_engine *e, uint64_t n, const uint32_t *q, const uint32_t *t,
                             const int32_t *diag, int32_t *score_out);
/* gapped DP (E5) for n pairs.  mode 0: forward (score, qend, tend); mode 1: reversed query (score);
 * mode 2: start pass on reverse(q[0..qend_in]) x reverse(t[0..tend_in]) (score, qend', tend').
 * Pairs may be in any order (the engine groups them by query internally). */
int uc_engine_sw_batch(uc_engine *e, int mode, uint64_t n, const uint32_t *q, const uint32_t *t,
                       const int32_t *qend_in, const int32_t *tend_in,
                       int32_t *score_out, int32_t *qend_out, int32_t *tend_out);

#ifdef __cplusplus
}
#endif
#endif
