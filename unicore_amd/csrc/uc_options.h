// uc_options.h — clustering parameters and the Foldseek-style flag parser.
// The reference forwards `-c/--cluster-options` as an opaque whitespace-split string
// (/root/reference/src/modules/cluster.rs:35,49; default "-c 0.8", src/util/arg_parser.rs:238-239);
// this is the only channel for clustering parameters, so the engine parses Foldseek's flag names.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace uc {

constexpr int A = 21;    // alphabet incl. X
constexpr int KA = 20;   // k-mer alphabet
constexpr int K = 6;     // k-mer size
constexpr uint32_t KSPACE = 64000000u;  // 20^6

struct Params {
    // matrices (loaded from data files, used verbatim)
    int8_t S3[A * A];
    int8_t SA[A * A];
    std::string mat3di_path, mataa_path;
    // Optional rule (default off = matrices used verbatim): rescale a loaded matrix the way MMseqs2's SubstitutionMatrix does,
    // score' = round(bit_factor x log2 odds) = round(bit_factor x lambda x score / ln 2), lambda from the file's "# Lambda" header line
    // (ln 2 / 2 = half-bit units if the file has none).  Foldseek is believed to use 2.1 for 3Di and 1.4 for AA (EXT-UNVERIFIED).
    double bit_factor_3di = 0.0, bit_factor_aa = 0.0;
    // prefilter
    std::string pattern = "1101010011";
    int koff[K] = {0, 1, 3, 5, 8, 9};
    int span = 10;
    float sensitivity = 4.0f;
    int kmer_thr = -1;          // -1: derive from sensitivity
    int min_diag_hits = 2;
    int min_ungapped = 15;
    int max_seqs = 300;
    // gapped alignment
    int gap_open = 10, gap_ext = 1;
    int rev_correction = 1;
    int want_tb = 0;            // 1: traceback statistics (alnlen, idents, gaps) for every accepted pair (search path)
    int sym_dedup = 1;          // 1: mutual hits (q,t)/(t,q) share one forward and one reversed-query DP (needs symmetric matrices)
    bool mat_symmetric = false; // set by finalize_params
    int sw_pk = 1;              // 1: packed 16-bit DP kernel for queries <= 1024 rows (int32 re-run when flagged)
    double evalue = 0.01, lambda = 0.34657359027997264, Kconst = 0.1;
    float cov = 0.8f;
    int cov_mode = 0;
    float min_seq_id = 0.0f;
    // optional rules, default off (spec UC-1/B, UC-1/E)
    int comp_bias = 0;                      // --comp-bias-corr 1: compositional bias on the ungapped score
    int comp_bias_milli = 1000;             // --comp-bias-corr-scale F, in thousandths (exact integer arithmetic on both sides)
    std::string min_score_table_path;       // --min-score-table FILE: one integer per sequence replaces the Karlin-Altschul threshold (plain step only)
    std::vector<int32_t> min_score_table;
    int len_gate = 0;                       // --length-gate 1 (rule UC-1/L, optional): pairs whose lengths alone rule the coverage threshold out are not aligned
    // clustering
    int cluster_mode = 0;
    int cluster_steps = 1;      // > 1: cascade (E8) — rounds on representatives with rising sensitivity, merged at the end
    int linclust = 0;           // 1: linear-time pre-step (E8a) before the clustering rounds
    int kmer_per_seq = 20;      // k-mers every sequence keeps in the pre-step
    bool kmer_thr_explicit = false;   // --k-score given: every cascade round uses it
    bool single_step = true;
    bool single_step_given = false, cluster_steps_given = false, linclust_given = false;
    bool mat3di_synthetic = false;    // the stand-in matrix is in use (UC_ALLOW_SYNTHETIC=1)
    // multi-GPU: devices uc_cluster spreads over (0 = all visible) and the target-shard count of the
    // Q x T grid (0 = one target shard per GPU, the north-star layout)
    int num_gpus = 1;
    int target_shards = 0;
    // runtime
    int threads = 1;
    int verbosity = 3;
};

// letters ACDEFGHIKLMNPQRSTVWY -> 0..19, everything else -> 20
int letter_code(char c);
void load_matrix(const std::string &path, int8_t out[A * A]);
// parse `opts` (Foldseek flag names) into p; throws Error(UC_ERR_ARGS) on unknown flags / bad values
void parse_cluster_options(const std::string &opts, Params &p);
// 1 = flag with a value, 2 = switch with an optional 0/1, -1 = unknown
int option_arity(const std::string &flag);
// resolve data files + derived values (pattern offsets, kmer_thr from sensitivity); loads matrices
void finalize_params(Params &p, const std::string &data_dir);
std::string default_data_dir();
// smallest integer score S with K * lq * db_residues * exp(-lambda*S) <= evalue
// sensitivity -> k-mer threshold (the rule of finalize_params)
int kmer_thr_for(const Params &p, double sensitivity);
int32_t min_score_for(const Params &p, int lq, uint64_t db_residues);

}  // namespace uc
