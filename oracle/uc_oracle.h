/* uc_oracle.h — CPU ORACLE for the `unicore cluster` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this code.
 * The product (unicore_amd/csrc, libunicore_cluster.so) never includes, links or executes anything
 * in oracle/.
 *
 * PARITY UNPINNED.  The reference (steineggerlab/unicore v1.1.1) performs this path by spawning the
 * third-party binary `foldseek` (src/modules/cluster.rs:45-76: `cluster`, `createtsv`, `rmdb`).
 * Foldseek (any version >= 10, README.md:36,237) is un-vendored, absent from /root/reference and from
 * this image, and the reference holds no test, golden vector or fixture for this path
 * (src/main.rs:68-81 are its only tests).  This file therefore restates the PUBLISHED algorithm
 * (van Kempen et al. 2024 Foldseek; Steinegger & Soeding 2017 MMseqs2; Farrar 2007 / Zhao 2013 SSW;
 * SURVEY.md Appendix A) with every tie-break frozen below, and is pinned only by (i) hand-computed
 * known-answer tests (tests/test_oracle_kat.py), (ii) the clust.tsv invariants the reference's
 * consumer needs (src/modules/profile.rs:50-55,79-84), (iii) committed oracle-generated fixtures
 * (tests/golden/, generator script alongside).
 *
 * ---------------------------------------------------------------------------------------------
 * FROZEN SPEC "UC-1"  (what both this oracle and the HIP path implement, bit-for-bit)
 *
 * Alphabet   letters ACDEFGHIKLMNPQRSTVWY -> 0..19, anything else -> 20 (X).  Both tracks (AA and
 *            3Di) use the same table (3Di states are written with the 20 AA letters).
 * Matrices   S3 (3Di) and SA (AA): 21x21 int8 loaded from MMseqs-style .out files, used verbatim.
 * E1 index   k = 6 spaced k-mers over the 3Di track, pattern P (default "1101010011", span 10):
 *            for target t, position j in [0, Lt-span]: letters c_m = t3[j+off_m]; skipped if any c_m
 *            is X; value v = sum c_m * 20^m.  Index = CSR v -> [(t, j)] ordered by (t, j).
 * E2 match   for query q, position i: similar set Sim = { v' : sum_m S3[c_m][c'_m] >= kmer_thr };
 *            every index entry (t, j) of every v' in Sim is a hit on diagonal d = i - j.
 *            For each (q,t): cnt(d) = number of hits on d;  d* = argmax cnt (tie: smallest d);
 *            (q,t,d*) is a candidate iff cnt(d*) >= min_diag_hits (2: the double-hit rule).
 * E3 ungapped  over i in [max(0,d), min(Lq, Lt+d)): run = max(0, run + S3[q3[i]][t3[i-d]]),
 *            best = max(best, run);  score = min(best, 255).
 * E4 select  keep score >= min_ungapped (15); order by (score desc, t asc); truncate to max_seqs.
 * E5 gapped  cell score s(i,j) = S3[q3[i]][t3[j]] + SA[qa[i]][ta[j]];  affine gaps: first gap
 *            residue costs `open`, each further `ext`:
 *               E(i,j) = max(E(i,j-1) - ext, H(i,j-1) - open)      (gap consuming target)
 *               F(i,j) = max(F(i-1,j) - ext, H(i-1,j) - open)      (gap consuming query)
 *               H(i,j) = max(0, H(i-1,j-1) + s(i,j), E(i,j), F(i,j)),   boundaries H=0.
 *            fwd pass : score = max H; tEnd = smallest j with a cell == score; qEnd = smallest i
 *                       with H(i,tEnd) == score.
 *            rev pass : score_rev = max H of (reversed query) x target   (composition correction,
 *                       SURVEY.md A.3);  corrected = score - score_rev   (if rev_correction).
 *                       UC-1.1: run only if score >= min_score(Lq); otherwise score_rev = 0 (the pair
 *                       cannot pass accept-1 because corrected <= score), outputs are unchanged.
 *            accept-1 : corrected >= min_score(Lq) where min_score(L) = smallest integer S with
 *                       K * L * db_residues * exp(-lambda * S) <= evalue.
 *            start pass (only if accept-1): same DP on reverse(q[0..qEnd]) x reverse(t[0..tEnd]) with
 *                       the same tie-break -> (qe', te');  qStart = qEnd - qe', tStart = tEnd - te'.
 * E6 accept  qcov = (float)(qEnd-qStart+1)/(float)Lq, tcov likewise; cov_mode 0: both >= cov;
 *            1: tcov; 2: qcov.  If min_seq_id > 0: seqId = identities/alnLen from the traceback
 *            (diag preferred over F over E; E/F leave the gap as soon as they can) must be >= it.
 * E7 set-cover  undirected graph on accepted pairs (plus self loops); repeat: pick the unassigned
 *            node with most unassigned neighbours (itself included; tie: smallest id) as
 *            representative and assign all its unassigned neighbours to it.
 * E8 cascade (only on request, steps > 1): `steps` rounds of E1-E7, round r on the representatives of round
 *            r-1, k-mer threshold from sensitivity s_r = 1 + (s-1) r/(steps-1); final representative =
 *            representative of the representative (mergeclusters).  No linclust pre-step.
 * E8a linclust-style pre-step (only on request, --linclust 1): see uco_linclust_pairs.
 * S  search (query DB vs target DB): E2-E6 per query against the index of the whole
 *            target DB, E-value with residues(target DB); traceback statistics (alnlen, idents, gaps) for every
 *            accepted pair; BLAST-tab rows per query by (corrected desc, target asc) - see uco_write_m8.
 * E9 TSV     clusters by ascending representative id; rows "rep\tmember": representative first,
 *            then the other members by ascending id.
 * ---------------------------------------------------------------------------------------------
 */
#ifndef UC_ORACLE_H
#define UC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UCO_A 21          /* alphabet incl. X */
#define UCO_KA 20         /* k-mer alphabet (X excluded) */
#define UCO_K 6
#define UCO_MAXSPAN 32

typedef struct uco_params {
    int8_t S3[UCO_A * UCO_A];
    int8_t SA[UCO_A * UCO_A];
    char pattern[UCO_MAXSPAN + 1];
    int kmer_thr;
    int min_diag_hits;
    int min_ungapped;
    int max_seqs;
    int gap_open, gap_ext;
    int rev_correction;
    double evalue, lambda, K;
    float cov;
    int cov_mode;
    float min_seq_id;
    int want_tb;                  /* 1: traceback statistics for every accepted pair (search / convertalis) */
    /* ---- optional rules, default off (walking towards Foldseek once a binary can be diffed: INTEGRATION.md section D) ----
     * UC-1/B  compositional bias on the ungapped score (restates MMseqs2 SubstitutionMatrix::calcLocalAaBiasCorrection on the 3Di
     *         track): query position i gets bias_i = round(scale * (rowsum(q_i) / 20 - sum_{j in window(i), j != i} S3[q_i][q_j] / |window|)),
     *         window = [max(0, i - 20), min(L, i + 20)), uniform background over the 20 letters, scale = comp_bias_milli / 1000, computed
     *         in exact integer arithmetic (round half away from zero); E3 scores S3[q_i][t_j] + bias_i.  The k-mer stage is untouched.
     * UC-1/E  per-query gate: min_score_table[q] (one integer per sequence of the database) replaces the Karlin-Altschul threshold on the
     *         corrected score — the hook for a fitted per-query E-value model (Foldseek predicts mu / lambda per query); plain step only. */
    int comp_bias_milli;          /* 0 = off */
    const int32_t *min_score_table;
    /* UC-1/L  length gate before E5 (restates MMseqs2 Util::canBeCovered as its alignment modules call it before aligning a pair; that
     *         Foldseek's structurealign does the same is EXT-UNVERIFIED): a pair whose LENGTHS alone rule the coverage threshold out is not
     *         aligned at all - cov_mode 0: Lq / Lt >= cov and Lt / Lq >= cov; 1 (target coverage): Lq / Lt >= cov; 2 (query coverage):
     *         Lt / Lq >= cov, float division.  Its record stays all-zero (never accepted), it counts neither as an alignment nor in the cell
     *         counters.  A heuristic, not a bound: gaps can stretch a short sequence over 80 % of a longer one. */
    int len_gate;                 /* 0 = off */
} uco_params;

typedef struct uco_db {           /* sequences as codes 0..20, concatenated, no padding */
    uint32_t n;
    uint64_t *off;                /* n+1 */
    uint8_t *s3, *sa;
    char **names;                 /* may be NULL */
} uco_db;

typedef struct uco_index {        /* CSR k-mer index over targets [tbegin,tend) */
    uint32_t *koff;               /* 20^6 + 1 */
    uint32_t *ent_seq;
    uint16_t *ent_pos;
    uint64_t n_entries;
} uco_index;

typedef struct uco_hit { uint32_t t; int32_t score; int32_t diag; } uco_hit;

typedef struct uco_aln {
    int32_t score, score_rev, corrected;
    int32_t qstart, qend, tstart, tend;   /* valid iff pass_evalue */
    int32_t aln_len, idents;              /* valid iff computed (min_seq_id > 0) */
    int32_t pass_evalue, accepted;
    int32_t gap_opens;                    /* with aln_len/idents: number of gaps on the traceback */
} uco_aln;

typedef struct uco_counts {
    uint64_t n_sim_kmers, n_kmer_hits, n_candidates, n_prefilter_hits, n_alignments, n_edges, n_clusters;
    uint64_t cells_fwd, cells_rev, cells_start;
} uco_counts;

/* rule UC-1/L; 1 when the rule is off */
int  uco_can_be_covered(const uco_params *p, int lq, int lt);
int  uco_letter_code(char c);
void uco_params_default(uco_params *p);
int  uco_load_matrix(const char *path, int8_t out[UCO_A * UCO_A]);
/* optional rule UC-1/M: score' = round(bit_factor * lambda * score / ln 2); lambda from the file header (0 = half-bit units) */
int  uco_rescale_matrix(int8_t m[UCO_A * UCO_A], double bit_factor, double lambda);
double uco_matrix_header_lambda(const char *path);

int  uco_db_read(const char *prefix, uco_db *db);     /* <prefix>, <prefix>_ss, <prefix>_h (+.index) */
void uco_db_free(uco_db *db);

int  uco_pattern_offsets(const char *pattern, int off[UCO_K]);   /* returns span or -1 */
int  uco_index_build(const uco_db *db, uint32_t tbegin, uint32_t tend, const uco_params *p, uco_index *ix);
void uco_index_free(uco_index *ix);

/* similar k-mers of the k-mer with letters c[0..5]; out may be NULL (count only); returns count */
size_t uco_similar_kmers(const int8_t S3[UCO_A * UCO_A], const uint8_t c[UCO_K], int thr, uint32_t *out, size_t cap);

int32_t uco_ungapped(const uint8_t *q3, int lq, const uint8_t *t3, int lt, int diag, const int8_t S3[UCO_A * UCO_A]);
/* rule UC-1/B: per-position bias of a query (out[lq]); uco_ungapped with it added per query position (qbias may be NULL) */
void uco_comp_bias(const uint8_t *q3, int lq, const int8_t S3[UCO_A * UCO_A], int scale_milli, int8_t *out);
int32_t uco_ungapped_bias(const uint8_t *q3, int lq, const uint8_t *t3, int lt, int diag, const int8_t S3[UCO_A * UCO_A], const int8_t *qbias);

/* E2+E3+E4 for one query; hits must hold max_seqs; cand (optional, cap ncand_cap) receives the
   pre-selection candidates (t, ungapped score, diag) in (t asc) order */
int  uco_prefilter_query(const uco_db *db, const uco_index *ix, uint32_t q, const uco_params *p,
                         uco_hit *hits, uco_counts *cnt);

/* gapped DP, score + end (fwd tie-break). rev_q / rev_t read the given prefix backwards. */
void uco_sw(const uint8_t *q3, const uint8_t *qa, int lq, int rev_q,
            const uint8_t *t3, const uint8_t *ta, int lt, int rev_t,
            const uco_params *p, int32_t *score, int32_t *qend, int32_t *tend);

int32_t uco_min_score(const uco_params *p, int lq, uint64_t db_residues);
int32_t uco_min_score_q(const uco_params *p, uint32_t q, int lq, uint64_t db_residues);   /* rule UC-1/E: the table entry of q if a table is set */
void uco_align_pair(const uco_db *db, uint32_t q, uint32_t t, const uco_params *p, int32_t min_score, uco_aln *out);

/* greedy set cover; edges are (a,b) pairs, any direction, duplicates allowed; assign[i] = representative id */
int  uco_setcover(uint32_t n, const uint32_t *edges, uint64_t n_edges, uint32_t *assign);

/* full pipeline (OpenMP over queries when compiled with -fopenmp); assign[n]; optional dumps:
   hits_out/hit_cnt_out: n*max_seqs hits and n counts;  aln_out aligned with hits_out */
int  uco_cluster(const uco_db *db, const uco_params *p, int threads, uint32_t *assign, uco_counts *cnt,
                 uco_hit *hits_out, uint32_t *hit_cnt_out, uco_aln *aln_out);

/* E8 cascade (spec UC-1 E8; Foldseek's default clustering workflow, SURVEY.md A.6, EXT-UNVERIFIED): `steps` rounds
   of the full pipeline, round r on the representatives of round r-1 with k-mer threshold thr[r] (sensitivity
   rising to the target: s_r = 1 + (s - 1) * r / (steps - 1), thr = the same sensitivity -> threshold rule as the
   single step); every round sees its own database size (E-value); the final representative of a sequence is
   the representative of its representative ... (== mergeclusters).  No linear-time (linclust) pre-step.
   steps == 1 is uco_cluster.  cnt (optional) receives the sums over the rounds. */
int  uco_cluster_cascade(const uco_db *db, const uco_params *p, int steps, const int *thr, int threads,
                         uint32_t *assign, uco_counts *cnt, uint32_t *round_sizes);

/* E8a linear-time pre-step (spec UC-1 E8a; restates Linclust, Steinegger & Soeding 2018, EXT-UNVERIFIED for Foldseek's
   parameters): every sequence keeps the m k-mers (the spaced 3Di k-mers of E1) with the smallest (hash, position),
   hash = SplitMix64 finaliser of the k-mer value; sequences that kept the same k-mer form a group; the centre of a
   group is its longest sequence (ties: smallest id); every (centre, member) pair - unique over all groups - goes
   through E5/E6 with the centre as query; accepted pairs are the edges of an E7 set cover over all sequences.
   pairs_out (optional): malloc'ed (centre, member) list sorted by (centre, member), caller frees. */
int  uco_linclust_pairs(const uco_db *db, const uco_params *p, int m, uint32_t **pairs_out, uint64_t *n_pairs);
int  uco_cluster_linclust(const uco_db *db, const uco_params *p, int m, int threads, uint32_t *assign, uco_counts *cnt);
/* the clustering workflow: optional E8a pre-step, then `steps` cascade rounds (E8) on its representatives */
int  uco_cluster_workflow(const uco_db *db, const uco_params *p, int linclust_m, int steps, const int *thr, int threads,
                          uint32_t *assign, uco_counts *cnt, uint32_t *round_sizes /* steps + 1 */);

int  uco_write_tsv(const char *path, const uco_db *db, const uint32_t *assign);

/* ---- search path (`foldseek search` + `convertalis`, reference src/modules/search.rs:44-61) ----
   spec UC-1 S: every query of qdb against the index of ALL of tdb through E2-E6 unchanged (E-value with the
   residue count of tdb), traceback statistics for every accepted pair (want_tb is forced on).  hits_out /
   aln_out: nq * max_seqs records, hit_cnt_out: nq counts. */
int  uco_search(const uco_db *qdb, const uco_db *tdb, const uco_params *p, int threads,
                uco_hit *hits_out, uint32_t *hit_cnt_out, uco_aln *aln_out, uco_counts *cnt);
/* BLAST-tab rows (== convertalis default columns): query target fident alnlen mismatch gapopen qstart qend
   tstart tend evalue bits; accepted pairs only, queries in DB order, per query (corrected score desc, target asc);
   positions 1-based; fident %.3f = idents/alnlen; mismatch = aligned pairs - idents with aligned pairs =
   qspan + tspan - alnlen; evalue %.3E = K*Lq*residues(tdb)*exp(-lambda*corrected); bits %d = trunc((lambda*corrected
   - ln K)/ln 2) */
int  uco_write_m8(const char *path, const uco_db *qdb, const uco_db *tdb, const uco_params *p,
                  const uco_hit *hits, const uint32_t *hit_cnt, const uco_aln *aln);

/* CPU-baseline helper (bench.py cpu_baseline leg): E2-E6 for the listed queries against a prebuilt index;
   returns the number of gapped alignments done; seconds[0] = prefilter wall, seconds[1] = alignment wall */
uint64_t uco_sample_run(const uco_db *db, const uco_index *ix, const uco_params *p, int threads,
                        const uint32_t *queries, uint32_t n_queries, double seconds[2]);

/* traceback statistics of the box [qs..qe] x [ts..te] (spec E6) */
void uco_traceback(const uco_db *db, uint32_t q, uint32_t t, const uco_params *p, int qs, int qe, int ts, int te,
                   int32_t *aln_len, int32_t *idents, int32_t *gap_opens);

/* ---- vectorised CPU leg (uc_simd.c; bench.py cpu_baseline "kind": "simd") -------------------------------------------
   E5/E6 of one query against its hit list with inter-sequence AVX2 Smith-Waterman: results identical to uco_align_pair */
void uco_simd_align_query(const uco_db *db, uint32_t q, const uint32_t *targets, uint32_t nh, const uco_params *p, int32_t min_score,
                          uco_aln *out);
/* uco_sample_run with that gapped stage; hit_out / cnt_out / aln_out (optional: n_queries * max_seqs, n_queries) */
uint64_t uco_simd_sample_run(const uco_db *db, const uco_index *ix, const uco_params *p, int threads,
                             const uint32_t *queries, uint32_t n_queries, double seconds[2],
                             uco_hit *hit_out, uint32_t *cnt_out, uco_aln *aln_out);
/* ... plus the prefilter stage counters of these queries added to *pc (may be NULL) */
uint64_t uco_simd_sample_run_counts(const uco_db *db, const uco_index *ix, const uco_params *p, int threads,
                                    const uint32_t *queries, uint32_t n_queries, double seconds[2],
                                    uco_hit *hit_out, uint32_t *cnt_out, uco_aln *aln_out, uco_counts *pc);
/* E5/E6 of a pair list sorted by query (pre-step pairs), one SIMD query group at a time, groups in parallel; out[np] in list order */
void uco_simd_align_pairs(const uco_db *db, const uint32_t *pairs, uint64_t np, const uco_params *p, int threads, uco_aln *out);
/* DP cells since the last call: out[0] = rows x columns of the problems handed to the SIMD kernel (forward, reversed and start
 * passes), out[1] = cells the 16-lane batches swept including lane padding; resets both */
void uco_simd_cells(uint64_t out[2]);

#ifdef __cplusplus
}
#endif
#endif
