// uc_options.cpp — Foldseek-style option parsing, matrix loading, derived parameters.
#include "uc_options.h"

#include <dlfcn.h>
#include <sys/stat.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "uc_common.h"

namespace uc {

int g_verbosity = 3;
static thread_local std::string t_last_error;
void set_last_error(const std::string &m) { t_last_error = m; }
const char *last_error_cstr() { return t_last_error.c_str(); }

int letter_code(char c) {
    switch (c >= 'a' && c <= 'z' ? c - 32 : c) {
        case 'A': return 0;  case 'C': return 1;  case 'D': return 2;  case 'E': return 3;  case 'F': return 4;
        case 'G': return 5;  case 'H': return 6;  case 'I': return 7;  case 'K': return 8;  case 'L': return 9;
        case 'M': return 10; case 'N': return 11; case 'P': return 12; case 'Q': return 13; case 'R': return 14;
        case 'S': return 15; case 'T': return 16; case 'V': return 17; case 'W': return 18; case 'Y': return 19;
        default: return 20;
    }
}

static bool is_matrix_letter(const std::string &tok) {
    return tok.size() == 1 && (letter_code(tok[0]) < 20 || tok[0] == 'X' || tok[0] == 'x');
}

// MMseqs2 / Foldseek matrix files carry their scale in comment lines: "# Lambda     (precision=N):" followed by "# <value>" (and the
// same for "# Background").  Returns the lambda of the file, 0 if it states none.
double matrix_header_lambda(const std::string &path) {
    std::ifstream f(path);
    std::string line;
    bool next = false;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] != '#') { if (!line.empty()) break; continue; }
        if (next) {
            const double v = std::strtod(line.c_str() + 1, nullptr);
            return v > 0 && v < 10 ? v : 0.0;
        }
        if (line.find("Lambda") != std::string::npos) {
            const size_t c = line.find(':');      // value on the same line ("# Lambda: 0.3466") or on the next comment line
            if (c != std::string::npos) { const double v = std::strtod(line.c_str() + c + 1, nullptr); if (v > 0 && v < 10) return v; }
            next = true;
        }
    }
    return 0.0;
}

// score' = round(bit_factor x lambda x score / ln 2): the integer a log-odds matrix in units of 1 / lambda nats becomes at
// bit_factor units per bit.  Values must stay inside the DP kernels' biased-byte range.
void rescale_matrix(int8_t m[A * A], double bit_factor, double lambda, const std::string &what) {
    if (!(bit_factor > 0)) return;
    if (!(lambda > 0)) lambda = std::log(2.0) / 2.0;
    for (int i = 0; i < A * A; i++) {
        const long v = std::lround(bit_factor * lambda * (double)m[i] / std::log(2.0));
        if (v < -48 || v > 48) fail(UC_ERR_ARGS, "%s: rescaled value %ld outside [-48,48] (bit factor %g)", what.c_str(), v, bit_factor);
        m[i] = (int8_t)v;
    }
}

void load_matrix(const std::string &path, int8_t out[A * A]) {
    std::ifstream f(path);
    if (!f) fail(UC_ERR_IO, "cannot open substitution matrix %s", path.c_str());
    for (int i = 0; i < A * A; i++) out[i] = -1;
    std::string line;
    std::vector<int> cols;
    int rows = 0;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string tok;
        if (!(ss >> tok) || tok[0] == '#') continue;
        if (cols.empty()) {
            do cols.push_back(is_matrix_letter(tok) ? letter_code(tok[0]) : -1);
            while (ss >> tok);
            continue;
        }
        if (tok.size() != 1) continue;
        int row = is_matrix_letter(tok) ? letter_code(tok[0]) : -1;
        for (size_t c = 0; c < cols.size(); c++) {
            long v;
            if (!(ss >> v)) break;
            if (row < 0 || cols[c] < 0) continue;
            // 48 keeps (S3 + SA + gap_open) inside the biased-byte query profile of the DP kernel
            if (v < -48 || v > 48) fail(UC_ERR_ARGS, "matrix %s: value %ld outside [-48,48]", path.c_str(), v);
            out[row * A + cols[c]] = (int8_t)v;
        }
        if (row >= 0) rows++;
    }
    if (rows < 20) fail(UC_ERR_ARGS, "matrix %s: expected 20 letter rows, found %d", path.c_str(), rows);
}

static double to_double(const std::string &flag, const std::string &v) {
    char *end = nullptr;
    double d = std::strtod(v.c_str(), &end);
    if (end == v.c_str() || *end) fail(UC_ERR_ARGS, "option %s: '%s' is not a number", flag.c_str(), v.c_str());
    return d;
}
static int to_int(const std::string &flag, const std::string &v) {
    char *end = nullptr;
    long d = std::strtol(v.c_str(), &end, 10);
    if (end == v.c_str() || *end) fail(UC_ERR_ARGS, "option %s: '%s' is not an integer", flag.c_str(), v.c_str());
    return (int)d;
}

// 1 = takes a value, 2 = MMseqs-style switch with an optional 0/1, -1 = not a flag of this engine.
// The shim uses the same table to split Foldseek's argv into positionals and options (flags may come anywhere).
int option_arity(const std::string &f) {
    static const char *const valued[] = {
        "-c", "--cov-mode", "--min-seq-id", "-e", "-s", "--max-seqs", "--k-score", "--min-ungapped-score", "--min-diag-hits",
        "--gap-open", "--gap-extend", "--spaced-kmer-pattern", "--rev-correction", "--linclust", "--kmer-per-seq", "--sym-dedup",
        "--sw-kernel", "--evalue-lambda", "--evalue-k", "--mat3di", "--mat-aa", "--cluster-mode", "--cluster-steps",
        "--alignment-type", "--alignment-mode", "--threads", "-v", "--remove-tmp-files", "--db-load-mode", "--compressed",
        "--gpus", "--target-shards", "--mat-bit-factor-3di", "--mat-bit-factor-aa", "--comp-bias-corr", "--comp-bias-corr-scale", "--min-score-table", "--length-gate"};
    for (const char *v : valued) if (f == v) return 1;
    if (f == "--single-step-clustering") return 2;
    return -1;
}

void parse_cluster_options(const std::string &opts, Params &p) {
    std::istringstream ss(opts);
    std::vector<std::string> tok;
    for (std::string t; ss >> t;) tok.push_back(t);
    for (size_t i = 0; i < tok.size(); i++) {
        const std::string &f = tok[i];
        auto value = [&]() -> const std::string & {
            if (i + 1 >= tok.size()) fail(UC_ERR_ARGS, "option %s needs a value", f.c_str());
            return tok[++i];
        };
        auto opt_bool = [&]() -> bool {  // MMseqs-style switches accept an optional 0/1
            if (i + 1 < tok.size() && (tok[i + 1] == "0" || tok[i + 1] == "1")) return tok[++i] == "1";
            return true;
        };
        if (f == "-c") { p.cov = (float)to_double(f, value()); if (p.cov < 0.f || p.cov > 1.f) fail(UC_ERR_ARGS, "-c must be in [0,1]"); }
        else if (f == "--cov-mode") { p.cov_mode = to_int(f, value()); if (p.cov_mode < 0 || p.cov_mode > 2) fail(UC_ERR_ARGS, "--cov-mode %d unsupported (0,1,2)", p.cov_mode); }
        else if (f == "--min-seq-id") { p.min_seq_id = (float)to_double(f, value()); if (p.min_seq_id < 0.f || p.min_seq_id > 1.f) fail(UC_ERR_ARGS, "--min-seq-id must be in [0,1]"); }
        else if (f == "-e") { p.evalue = to_double(f, value()); if (!(p.evalue > 0)) fail(UC_ERR_ARGS, "-e must be > 0"); }
        else if (f == "-s") { p.sensitivity = (float)to_double(f, value()); }
        else if (f == "--max-seqs") { p.max_seqs = to_int(f, value()); if (p.max_seqs < 1 || p.max_seqs > 65535) fail(UC_ERR_ARGS, "--max-seqs must be in [1,65535]"); }
        else if (f == "--k-score") { p.kmer_thr = to_int(f, value()); }
        else if (f == "--min-ungapped-score") { p.min_ungapped = to_int(f, value()); }
        else if (f == "--min-diag-hits") { p.min_diag_hits = to_int(f, value()); if (p.min_diag_hits < 1) fail(UC_ERR_ARGS, "--min-diag-hits must be >= 1"); }
        else if (f == "--gap-open") { p.gap_open = to_int(f, value()); if (p.gap_open < 1 || p.gap_open > 31) fail(UC_ERR_ARGS, "--gap-open must be in [1,31]"); }
        else if (f == "--gap-extend") { p.gap_ext = to_int(f, value()); if (p.gap_ext < 0 || p.gap_ext > 31) fail(UC_ERR_ARGS, "--gap-extend must be in [0,31]"); }
        else if (f == "--spaced-kmer-pattern") { p.pattern = value(); }
        else if (f == "--rev-correction") { p.rev_correction = to_int(f, value()) != 0; }
        else if (f == "--linclust") { p.linclust = to_int(f, value()) != 0; p.linclust_given = true; }
        else if (f == "--kmer-per-seq") { p.kmer_per_seq = to_int(f, value()); if (p.kmer_per_seq < 1 || p.kmer_per_seq > 1000) fail(UC_ERR_ARGS, "--kmer-per-seq must be in [1,1000]"); }
        else if (f == "--sym-dedup") { p.sym_dedup = to_int(f, value()) != 0; }
        else if (f == "--sw-kernel") { const std::string &v = value(); if (v == "pk16") p.sw_pk = 1; else if (v == "i32") p.sw_pk = 0; else fail(UC_ERR_ARGS, "--sw-kernel must be pk16 or i32"); }
        else if (f == "--evalue-lambda") { p.lambda = to_double(f, value()); }
        else if (f == "--evalue-k") { p.Kconst = to_double(f, value()); }
        else if (f == "--mat3di") { p.mat3di_path = value(); }
        else if (f == "--mat-aa") { p.mataa_path = value(); }
        else if (f == "--mat-bit-factor-3di") { p.bit_factor_3di = to_double(f, value()); if (p.bit_factor_3di < 0 || p.bit_factor_3di > 16) fail(UC_ERR_ARGS, "--mat-bit-factor-3di must be in [0,16]"); }
        else if (f == "--mat-bit-factor-aa") { p.bit_factor_aa = to_double(f, value()); if (p.bit_factor_aa < 0 || p.bit_factor_aa > 16) fail(UC_ERR_ARGS, "--mat-bit-factor-aa must be in [0,16]"); }
        else if (f == "--comp-bias-corr") { p.comp_bias = to_int(f, value()) != 0; }
        else if (f == "--comp-bias-corr-scale") { const double v = to_double(f, value()); if (v < 0 || v > 8) fail(UC_ERR_ARGS, "--comp-bias-corr-scale must be in [0,8]"); p.comp_bias_milli = (int)std::lround(v * 1000.0); }
        else if (f == "--min-score-table") { p.min_score_table_path = value(); }
        else if (f == "--length-gate") { p.len_gate = to_int(f, value()) != 0; }
        else if (f == "--cluster-mode") { p.cluster_mode = to_int(f, value()); if (p.cluster_mode != 0) fail(UC_ERR_ARGS, "--cluster-mode %d unsupported (only 0 = greedy set cover)", p.cluster_mode); }
        else if (f == "--single-step-clustering") { p.single_step = opt_bool(); p.single_step_given = true; }
        else if (f == "--cluster-steps") { p.cluster_steps = to_int(f, value()); p.cluster_steps_given = true; }
        else if (f == "--gpus") { p.num_gpus = to_int(f, value()); if (p.num_gpus < 0 || p.num_gpus > 64) fail(UC_ERR_ARGS, "--gpus must be in [0,64] (0 = all visible)"); }
        else if (f == "--target-shards") { p.target_shards = to_int(f, value()); if (p.target_shards < 0 || p.target_shards > 64) fail(UC_ERR_ARGS, "--target-shards must be in [0,64] (0 = one shard per GPU)"); }
        else if (f == "--alignment-type") { int v = to_int(f, value()); if (v != 2) fail(UC_ERR_ARGS, "--alignment-type %d unsupported (only 2 = 3Di+AA)", v); }
        else if (f == "--alignment-mode") { int v = to_int(f, value()); if (v < 0 || v > 3) fail(UC_ERR_ARGS, "--alignment-mode %d unsupported", v); }
        else if (f == "--threads") { p.threads = to_int(f, value()); }
        else if (f == "-v") { p.verbosity = to_int(f, value()); }
        else if (f == "--remove-tmp-files" || f == "--db-load-mode" || f == "--compressed") { (void)value(); /* no effect on results */ }
        else fail(UC_ERR_ARGS, "unknown or unsupported cluster option '%s'", f.c_str());
    }
}

std::string default_data_dir() {
    Dl_info info;
    if (dladdr((void *)&default_data_dir, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t s = p.find_last_of('/');
        return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/data";
    }
    return "data";
}

static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

void finalize_params(Params &p, const std::string &data_dir_in) {
    std::string dd = data_dir_in.empty() ? default_data_dir() : data_dir_in;
    if (p.mat3di_path.empty()) {
        if (exists(dd + "/mat3di.out")) p.mat3di_path = dd + "/mat3di.out";
        else {
            // Foldseek's mat3di.out cannot be authored here; the shipped stand-in is a seeded random matrix.
            // Clustering real proteomes with it would look valid and mean nothing, so it needs an explicit opt-in.
            const char *allow = getenv("UC_ALLOW_SYNTHETIC");
            if (!allow || strcmp(allow, "1") != 0)
                fail(UC_ERR_ARGS, "no mat3di.out in %s: copy Foldseek's data/mat3di.out there (or pass --mat3di <path>). The shipped "
                                  "mat3di_synthetic.out is a seeded STAND-IN for tests and benchmarks; set UC_ALLOW_SYNTHETIC=1 to use it",
                     dd.c_str());
            p.mat3di_path = dd + "/mat3di_synthetic.out";
            p.mat3di_synthetic = true;
        }
    }
    if (p.mataa_path.empty()) p.mataa_path = dd + "/blosum62.out";
    load_matrix(p.mat3di_path, p.S3);
    load_matrix(p.mataa_path, p.SA);
    rescale_matrix(p.S3, p.bit_factor_3di, matrix_header_lambda(p.mat3di_path), p.mat3di_path);
    rescale_matrix(p.SA, p.bit_factor_aa, matrix_header_lambda(p.mataa_path), p.mataa_path);
    p.mat_symmetric = true;
    for (int a = 0; a < A; a++)
        for (int b = 0; b < a; b++)
            if (p.S3[a * A + b] != p.S3[b * A + a] || p.SA[a * A + b] != p.SA[b * A + a]) p.mat_symmetric = false;
    // pattern
    int n = 0;
    p.span = (int)p.pattern.size();
    if (p.span < K || p.span > 32) fail(UC_ERR_ARGS, "--spaced-kmer-pattern: span must be in [6,32]");
    for (int i = 0; i < p.span; i++) {
        if (p.pattern[i] == '1') { if (n == K) fail(UC_ERR_ARGS, "--spaced-kmer-pattern needs exactly 6 ones"); p.koff[n++] = i; }
        else if (p.pattern[i] != '0') fail(UC_ERR_ARGS, "--spaced-kmer-pattern must consist of 0/1");
    }
    if (n != K || p.pattern.front() != '1' || p.pattern.back() != '1') fail(UC_ERR_ARGS, "--spaced-kmer-pattern needs exactly 6 ones and 1 at both ends");
    p.kmer_thr_explicit = p.kmer_thr >= 0;
    if (p.kmer_thr < 0) p.kmer_thr = kmer_thr_for(p, p.sensitivity);
    // Workflow (E8).  Unicore forwards only "-c 0.8" (arg_parser.rs:238-239), so what runs is Foldseek's DEFAULT `cluster`
    // workflow: a linear-time pre-step + a 3-step cascade.  --single-step-clustering selects the plain
    // all-vs-all step; --cluster-steps / --linclust override the two halves individually.
    if (p.single_step_given && p.single_step) {
        p.cluster_steps = 1;
        if (!p.linclust_given) p.linclust = 0;
    } else {
        if (!p.cluster_steps_given) p.cluster_steps = 3;
        if (!p.linclust_given) p.linclust = 1;
    }
    if (p.cluster_steps < 1 || p.cluster_steps > 16) fail(UC_ERR_ARGS, "--cluster-steps must be in [1,16]");
    p.single_step = p.cluster_steps == 1 && !p.linclust;
    if (p.threads < 1) p.threads = 1;
    if (!p.min_score_table_path.empty()) {      // rule UC-1/E: per-query thresholds from a fitted model, whitespace-separated integers in database order
        if (!p.single_step) fail(UC_ERR_ARGS, "--min-score-table is indexed by database sequence: it needs --single-step-clustering (the rounds of the default workflow run on sub-databases)");
        FILE *f = fopen(p.min_score_table_path.c_str(), "r");
        if (!f) fail(UC_ERR_IO, "cannot open --min-score-table %s", p.min_score_table_path.c_str());
        p.min_score_table.clear();
        long v;
        int rc;
        while ((rc = fscanf(f, "%ld", &v)) == 1) {
            if (v < 1 || v > 30000) { fclose(f); fail(UC_ERR_ARGS, "--min-score-table: threshold %ld out of [1,30000]", v); }
            p.min_score_table.push_back((int32_t)v);
        }
        const bool junk = rc != EOF;
        fclose(f);
        if (junk) fail(UC_ERR_ARGS, "--min-score-table %s: not a list of integers", p.min_score_table_path.c_str());
    }
}

// sensitivity -> k-mer threshold: mean self score of a k-mer + 3 - 2*s  (data-driven stand-in for Foldseek's
// threshold table, SURVEY.md A.2 EXT-UNVERIFIED; --k-score overrides)
int kmer_thr_for(const Params &p, double sensitivity) {
    double diag = 0;
    for (int a = 0; a < KA; a++) diag += p.S3[a * A + a];
    return (int)std::lround(K * diag / KA + 3.0 - 2.0 * sensitivity);
}

int32_t min_score_for(const Params &p, int lq, uint64_t db_residues) {
    double scale = p.Kconst * (double)lq * (double)db_residues;
    int32_t s = 1;
    while (scale * std::exp(-p.lambda * (double)s) > p.evalue && s < (1 << 30)) s++;
    return s;
}

}  // namespace uc
