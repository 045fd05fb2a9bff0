/* uc_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see uc_oracle.h for the rules and spec UC-1).
 *
 * PARITY UNPINNED: restates the published Foldseek/MMseqs2 algorithm behind the three subprocess
 * calls of /root/reference/src/modules/cluster.rs:45-76; no reference golden vector exists.
 * Every function names the reference call site / published stage it stands for.
 */
#include "uc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ alphabet / matrices */

/* MMseqs2-style alphabetical letter table; X and everything unknown -> 20. */
int uco_letter_code(char c) {
    static const char LET[] = "ACDEFGHIKLMNPQRSTVWY";
    if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');
    const char *p = c ? strchr(LET, c) : NULL;
    return p ? (int)(p - LET) : 20;
}

void uco_params_default(uco_params *p) {
    memset(p, 0, sizeof *p);
    strcpy(p->pattern, "1101010011");
    p->kmer_thr = 22;
    p->min_diag_hits = 2;
    p->min_ungapped = 15;
    p->max_seqs = 300;
    p->gap_open = 10;
    p->gap_ext = 1;
    p->rev_correction = 1;
    p->evalue = 0.01;
    p->lambda = 0.34657359027997264; /* ln2/2: half-bit units */
    p->K = 0.1;
    p->cov = 0.8f;
    p->cov_mode = 0;
    p->min_seq_id = 0.0f;
}

/* MMseqs-style matrix text: '#' comments, a header row of column letters, then "L v v v ..." rows.
   Letters outside the 21-letter alphabet (B, Z, *) are ignored; missing X entries default to -1. */
int uco_load_matrix(const char *path, int8_t out[UCO_A * UCO_A]) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    for (int i = 0; i < UCO_A * UCO_A; i++) out[i] = -1;
    char line[4096];
    int cols[64], ncols = 0, seen_rows = 0;
    while (fgets(line, sizeof line, f)) {
        char *s = line;
        while (*s == ' ' || *s == '\t') s++;
        if (*s == '#' || *s == '\n' || *s == 0) continue;
        if (ncols == 0) { /* header */
            for (char *tok = strtok(s, " \t\r\n"); tok && ncols < 64; tok = strtok(NULL, " \t\r\n")) {
                int known = (strlen(tok) == 1) && (uco_letter_code(tok[0]) < 20 || tok[0] == 'X' || tok[0] == 'x');
                cols[ncols++] = known ? uco_letter_code(tok[0]) : -1;
            }
            continue;
        }
        char *tok = strtok(s, " \t\r\n");
        if (!tok || strlen(tok) != 1) continue;
        int known = uco_letter_code(tok[0]) < 20 || tok[0] == 'X' || tok[0] == 'x';
        int row = known ? uco_letter_code(tok[0]) : -1;
        for (int c = 0; c < ncols; c++) {
            tok = strtok(NULL, " \t\r\n");
            if (!tok) break;
            if (row >= 0 && cols[c] >= 0) {
                long v = strtol(tok, NULL, 10);
                if (v < -127 || v > 127) { fclose(f); return -2; }
                out[row * UCO_A + cols[c]] = (int8_t)v;
            }
        }
        if (row >= 0) seen_rows++;
    }
    fclose(f);
    return seen_rows >= 20 ? 0 : -3;
}

/* Optional rule UC-1/M (default off): the integer matrix at `bit_factor` units per bit, from a log-odds matrix whose unit is
 * 1 / lambda nats: score' = round(bit_factor * lambda * score / ln 2) (what MMseqs2's SubstitutionMatrix does with the bit factors
 * Foldseek passes for its two tracks: believed 2.1 for 3Di, 1.4 for AA — EXT-UNVERIFIED).  lambda <= 0 means half-bit units. */
int uco_rescale_matrix(int8_t m[UCO_A * UCO_A], double bit_factor, double lambda) {
    if (!(bit_factor > 0)) return 0;
    if (!(lambda > 0)) lambda = log(2.0) / 2.0;
    for (int i = 0; i < UCO_A * UCO_A; i++) {
        const long v = lround(bit_factor * lambda * (double)m[i] / log(2.0));
        if (v < -127 || v > 127) return -1;
        m[i] = (int8_t)v;
    }
    return 0;
}

/* the "# Lambda ..." header of an MMseqs2 / Foldseek matrix file (value on the same line after ':' or on the next comment line); 0 if none */
double uco_matrix_header_lambda(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) return 0.0;
    char line[4096];
    int next = 0;
    double out = 0.0;
    while (fgets(line, sizeof line, f)) {
        if (line[0] != '#') { if (line[0] != '\n' && line[0] != 0) break; continue; }
        if (next) { const double v = strtod(line + 1, NULL); if (v > 0 && v < 10) out = v; break; }
        if (strstr(line, "Lambda")) {
            const char *c = strchr(line, ':');
            if (c) { const double v = strtod(c + 1, NULL); if (v > 0 && v < 10) { out = v; break; } }
            next = 1;
        }
    }
    fclose(f);
    return out;
}

/* ------------------------------------------------------------------ DB reader
 * Format witness: /root/reference/src/seq/create_gene_specific_fasta.rs:9-36 (entries "TEXT\n\0",
 * three index-aligned files <db>, <db>_ss, <db>_h); index "key\toffset\tlength" (SURVEY.md App. B). */
static char *read_file(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    if (n > 0 && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); fclose(f); return NULL; }
    buf[n] = 0;
    fclose(f);
    *len = (size_t)n;
    return buf;
}
typedef struct { uint64_t key, off, len; } idx_ent;
static int cmp_idx(const void *a, const void *b) {
    const idx_ent *x = (const idx_ent *)a, *y = (const idx_ent *)b;
    return x->key < y->key ? -1 : x->key > y->key;
}
static idx_ent *read_index(const char *path, uint32_t *n) {
    size_t len; char *txt = read_file(path, &len);
    if (!txt) return NULL;
    size_t cap = 1024, cnt = 0;
    idx_ent *e = (idx_ent *)malloc(cap * sizeof *e);
    char *s = txt;
    while (*s) {
        char *end;
        uint64_t k = strtoull(s, &end, 10);
        if (end == s) break;
        uint64_t o = strtoull(end, &end, 10);
        uint64_t l = strtoull(end, &end, 10);
        if (cnt == cap) { cap *= 2; e = (idx_ent *)realloc(e, cap * sizeof *e); }
        e[cnt].key = k; e[cnt].off = o; e[cnt].len = l; cnt++;
        s = end;
        while (*s == '\n' || *s == '\r' || *s == ' ' || *s == '\t') s++;
    }
    free(txt);
    if (cnt) qsort(e, cnt, sizeof *e, cmp_idx);
    *n = (uint32_t)cnt;
    return e;
}

int uco_db_read(const char *prefix, uco_db *db) {
    memset(db, 0, sizeof *db);
    char path[4096];
    uint32_t na = 0, ns = 0, nh = 0;
    snprintf(path, sizeof path, "%s.index", prefix);     idx_ent *ia = read_index(path, &na);
    snprintf(path, sizeof path, "%s_ss.index", prefix);  idx_ent *is = read_index(path, &ns);
    snprintf(path, sizeof path, "%s_h.index", prefix);   idx_ent *ih = read_index(path, &nh);
    if (!ia || !is || !ih || na != ns || na != nh) { free(ia); free(is); free(ih); return -1; }
    size_t la, ls, lh;
    snprintf(path, sizeof path, "%s", prefix);     char *da = read_file(path, &la);
    snprintf(path, sizeof path, "%s_ss", prefix);  char *ds = read_file(path, &ls);
    snprintf(path, sizeof path, "%s_h", prefix);   char *dh = read_file(path, &lh);
    if (!da || !ds || !dh) { free(ia); free(is); free(ih); free(da); free(ds); free(dh); return -2; }
    db->n = na;
    db->off = (uint64_t *)malloc((na + 1) * sizeof(uint64_t));
    db->names = (char **)calloc(na, sizeof(char *));
    uint64_t tot = 0;
    for (uint32_t i = 0; i < na; i++) {
        if (ia[i].key != is[i].key || ia[i].key != ih[i].key) return -3;
        uint64_t l = ia[i].len >= 2 ? ia[i].len - 2 : 0;
        uint64_t l2 = is[i].len >= 2 ? is[i].len - 2 : 0;
        if (l != l2 || ia[i].off + l > la || is[i].off + l > ls) return -4;
        db->off[i] = tot; tot += l;
    }
    db->off[na] = tot;
    db->s3 = (uint8_t *)malloc(tot + 1); db->sa = (uint8_t *)malloc(tot + 1);
    for (uint32_t i = 0; i < na; i++) {
        uint64_t l = db->off[i + 1] - db->off[i];
        for (uint64_t k = 0; k < l; k++) {
            db->sa[db->off[i] + k] = (uint8_t)uco_letter_code(da[ia[i].off + k]);
            db->s3[db->off[i] + k] = (uint8_t)uco_letter_code(ds[is[i].off + k]);
        }
        const char *h = dh + ih[i].off;
        size_t hl = 0;
        while (ih[i].off + hl < lh && h[hl] && h[hl] != ' ' && h[hl] != '\t' && h[hl] != '\n') hl++;
        db->names[i] = (char *)malloc(hl + 1);
        memcpy(db->names[i], h, hl); db->names[i][hl] = 0;
    }
    free(ia); free(is); free(ih); free(da); free(ds); free(dh);
    return 0;
}

void uco_db_free(uco_db *db) {
    if (db->names) { for (uint32_t i = 0; i < db->n; i++) free(db->names[i]); free(db->names); }
    free(db->off); free(db->s3); free(db->sa);
    memset(db, 0, sizeof *db);
}

/* ------------------------------------------------------------------ E1: k-mer index */

int uco_pattern_offsets(const char *pattern, int off[UCO_K]) {
    int n = 0, span = (int)strlen(pattern);
    if (span > UCO_MAXSPAN) return -1;
    for (int i = 0; i < span; i++) {
        if (pattern[i] == '1') { if (n == UCO_K) return -1; off[n++] = i; }
        else if (pattern[i] != '0') return -1;
    }
    return n == UCO_K && pattern[0] == '1' && pattern[span - 1] == '1' ? span : -1;
}

#define KSPACE 64000000u /* 20^6 */

static inline int kmer_at(const uint8_t *s, const int off[UCO_K], uint32_t *v) {
    uint32_t x = 0, mul = 1;
    for (int m = 0; m < UCO_K; m++) {
        uint8_t c = s[off[m]];
        if (c >= UCO_KA) return 0;
        x += c * mul; mul *= UCO_KA;
    }
    *v = x;
    return 1;
}

/* MMseqs2 prefilter index table: CSR k-mer -> (seq, pos), X-containing k-mers skipped. */
int uco_index_build(const uco_db *db, uint32_t tbegin, uint32_t tend, const uco_params *p, uco_index *ix) {
    int off[UCO_K];
    int span = uco_pattern_offsets(p->pattern, off);
    if (span < 0) return -1;
    memset(ix, 0, sizeof *ix);
    ix->koff = (uint32_t *)calloc((size_t)KSPACE + 1, sizeof(uint32_t));
    if (!ix->koff) return -2;
    for (uint32_t t = tbegin; t < tend; t++) {
        const uint8_t *s = db->s3 + db->off[t];
        int64_t l = (int64_t)(db->off[t + 1] - db->off[t]);
        for (int64_t j = 0; j + span <= l && j <= 65535; j++) {
            uint32_t v;
            if (kmer_at(s + j, off, &v)) ix->koff[v + 1]++;
        }
    }
    for (uint32_t v = 0; v < KSPACE; v++) ix->koff[v + 1] += ix->koff[v];
    ix->n_entries = ix->koff[KSPACE];
    ix->ent_seq = (uint32_t *)malloc((ix->n_entries + 1) * sizeof(uint32_t));
    ix->ent_pos = (uint16_t *)malloc((ix->n_entries + 1) * sizeof(uint16_t));
    uint32_t *cur = (uint32_t *)malloc((size_t)KSPACE * sizeof(uint32_t));
    memcpy(cur, ix->koff, (size_t)KSPACE * sizeof(uint32_t));
    for (uint32_t t = tbegin; t < tend; t++) {
        const uint8_t *s = db->s3 + db->off[t];
        int64_t l = (int64_t)(db->off[t + 1] - db->off[t]);
        for (int64_t j = 0; j + span <= l && j <= 65535; j++) {
            uint32_t v;
            if (kmer_at(s + j, off, &v)) { uint32_t e = cur[v]++; ix->ent_seq[e] = t; ix->ent_pos[e] = (uint16_t)j; }
        }
    }
    free(cur);
    return 0;
}

void uco_index_free(uco_index *ix) { free(ix->koff); free(ix->ent_seq); free(ix->ent_pos); memset(ix, 0, sizeof *ix); }

/* ------------------------------------------------------------------ E2: similar k-mers (MMseqs2 KmerGenerator) */

typedef struct { uint32_t *v; size_t n, cap; int count_only; } kvec;
static void kvec_push(kvec *k, uint32_t x) {
    if (!k->count_only) {
        if (k->n == k->cap) { k->cap = k->cap ? k->cap * 2 : 256; k->v = (uint32_t *)realloc(k->v, k->cap * sizeof(uint32_t)); }
        k->v[k->n] = x;
    }
    k->n++;
}
static void sim_dfs(const int8_t *S3, const uint8_t *c, const int *restmax, int thr, int m, int partial,
                    uint32_t val, uint32_t mul, kvec *out) {
    if (m == UCO_K) { kvec_push(out, val); return; }
    const int8_t *row = S3 + c[m] * UCO_A;
    for (int b = 0; b < UCO_KA; b++) {
        int s = partial + row[b];
        if (s + restmax[m + 1] >= thr) sim_dfs(S3, c, restmax, thr, m + 1, s, val + (uint32_t)b * mul, mul * UCO_KA, out);
    }
}
static void similar_kmers(const int8_t *S3, const uint8_t c[UCO_K], int thr, kvec *out) {
    int restmax[UCO_K + 1];
    restmax[UCO_K] = 0;
    for (int m = UCO_K - 1; m >= 0; m--) {
        int best = -128;
        for (int b = 0; b < UCO_KA; b++) if (S3[c[m] * UCO_A + b] > best) best = S3[c[m] * UCO_A + b];
        restmax[m] = restmax[m + 1] + best;
    }
    if (restmax[0] < thr) return;
    sim_dfs(S3, c, restmax, thr, 0, 0, 0, 1, out);
}
size_t uco_similar_kmers(const int8_t S3[UCO_A * UCO_A], const uint8_t c[UCO_K], int thr, uint32_t *out, size_t cap) {
    kvec k = {0};
    k.count_only = (out == NULL);
    similar_kmers(S3, c, thr, &k);
    if (out) { memcpy(out, k.v, (k.n < cap ? k.n : cap) * sizeof(uint32_t)); free(k.v); }
    return k.n;
}

/* ------------------------------------------------------------------ E3: ungapped diagonal score
 * MMseqs2 UngappedAlignment on the 3Di track: Kadane along the whole diagonal,
 * saturating at 255. */
int32_t uco_ungapped_bias(const uint8_t *q3, int lq, const uint8_t *t3, int lt, int diag, const int8_t S3[UCO_A * UCO_A], const int8_t *qbias) {
    int i0 = diag > 0 ? diag : 0;
    int i1 = lq < lt + diag ? lq : lt + diag;
    int run = 0, best = 0;
    for (int i = i0; i < i1; i++) {
        run += S3[q3[i] * UCO_A + t3[i - diag]] + (qbias ? qbias[i] : 0);
        if (run < 0) run = 0;
        if (run > best) best = run;
    }
    return best > 255 ? 255 : best;
}
int32_t uco_ungapped(const uint8_t *q3, int lq, const uint8_t *t3, int lt, int diag, const int8_t S3[UCO_A * UCO_A]) {
    return uco_ungapped_bias(q3, lq, t3, lt, diag, S3, NULL);
}

/* rule UC-1/B (MMseqs2 SubstitutionMatrix::calcLocalAaBiasCorrection restated on the 3Di track, uniform background): exact integers.
   bias_i = round_half_away( scale/1000 * ( rowsum_i / 20 - sum_i / wl ) ) = round( scale * (rowsum_i * wl - 20 * sum_i) / (20000 * wl) ) */
void uco_comp_bias(const uint8_t *q3, int lq, const int8_t S3[UCO_A * UCO_A], int scale_milli, int8_t *out) {
    for (int i = 0; i < lq; i++) {
        const int lo = i - 20 > 0 ? i - 20 : 0, hi = i + 20 < lq ? i + 20 : lq, wl = hi - lo;
        const int8_t *row = S3 + q3[i] * UCO_A;
        int sum = 0, rowsum = 0;
        for (int j = lo; j < hi; j++) sum += row[q3[j]];
        sum -= row[q3[i]];
        for (int a = 0; a < UCO_KA; a++) rowsum += row[a];
        const long long num = (long long)scale_milli * ((long long)rowsum * wl - 20LL * sum), den = 20000LL * wl;
        long long b = num >= 0 ? (num + den / 2) / den : -((-num + den / 2) / den);
        out[i] = (int8_t)(b > 127 ? 127 : b < -128 ? -128 : b);
    }
}

/* ------------------------------------------------------------------ E2+E3+E4 for one query */
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
static int cmp_hit(const void *a, const void *b) {
    const uco_hit *x = (const uco_hit *)a, *y = (const uco_hit *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return x->t < y->t ? -1 : x->t > y->t;
}

int uco_prefilter_query(const uco_db *db, const uco_index *ix, uint32_t q, const uco_params *p,
                        uco_hit *hits, uco_counts *cnt) {
    int off[UCO_K];
    int span = uco_pattern_offsets(p->pattern, off);
    const uint8_t *q3 = db->s3 + db->off[q];
    int lq = (int)(db->off[q + 1] - db->off[q]);
    kvec sim = {0};
    uint64_t *hk = NULL; size_t nh = 0, hcap = 0;
    for (int i = 0; i + span <= lq && i <= 65535; i++) {
        uint8_t c[UCO_K]; int ok = 1;
        for (int m = 0; m < UCO_K; m++) { c[m] = q3[i + off[m]]; if (c[m] >= UCO_KA) ok = 0; }
        if (!ok) continue;
        sim.n = 0;
        similar_kmers(p->S3, c, p->kmer_thr, &sim);
        if (cnt) cnt->n_sim_kmers += sim.n;
        for (size_t k = 0; k < sim.n; k++) {
            uint32_t v = sim.v[k];
            for (uint32_t e = ix->koff[v]; e < ix->koff[v + 1]; e++) {
                if (nh == hcap) { hcap = hcap ? hcap * 2 : 4096; hk = (uint64_t *)realloc(hk, hcap * sizeof(uint64_t)); }
                int d = i - (int)ix->ent_pos[e];
                hk[nh++] = ((uint64_t)ix->ent_seq[e] << 32) | (uint32_t)(d + 65536);
            }
        }
    }
    free(sim.v);
    if (cnt) cnt->n_kmer_hits += nh;
    if (nh) qsort(hk, nh, sizeof(uint64_t), cmp_u64);   /* (qsort(NULL, 0, ..) is undefined: found by the UBSan job) */
    /* per target: diagonal with most hits (tie: smallest diagonal); double-hit rule */
    int8_t *qbias = NULL;                                   /* rule UC-1/B (off by default) */
    if (p->comp_bias_milli && lq > 0) { qbias = (int8_t *)malloc((size_t)lq); uco_comp_bias(q3, lq, p->S3, p->comp_bias_milli, qbias); }
    uco_hit *cand = NULL; size_t nc = 0, ccap = 0;
    size_t a = 0;
    while (a < nh) {
        uint32_t t = (uint32_t)(hk[a] >> 32);
        int best_cnt = 0, best_d = 0;
        size_t b = a;
        while (b < nh && (uint32_t)(hk[b] >> 32) == t) {
            size_t e = b;
            while (e < nh && hk[e] == hk[b]) e++;
            int c = (int)(e - b);
            if (c > best_cnt) { best_cnt = c; best_d = (int)(uint32_t)hk[b] - 65536; }
            b = e;
        }
        if (best_cnt >= p->min_diag_hits) {
            if (nc == ccap) { ccap = ccap ? ccap * 2 : 256; cand = (uco_hit *)realloc(cand, ccap * sizeof *cand); }
            const uint8_t *t3 = db->s3 + db->off[t];
            int lt = (int)(db->off[t + 1] - db->off[t]);
            cand[nc].t = t; cand[nc].diag = best_d;
            cand[nc].score = uco_ungapped_bias(q3, lq, t3, lt, best_d, p->S3, qbias);
            nc++;
        }
        a = b;
    }
    free(hk);
    free(qbias);
    if (cnt) cnt->n_candidates += nc;
    size_t kept = 0;
    for (size_t k = 0; k < nc; k++) if (cand[k].score >= p->min_ungapped) cand[kept++] = cand[k];
    if (kept) qsort(cand, kept, sizeof *cand, cmp_hit);
    if (kept > (size_t)p->max_seqs) kept = (size_t)p->max_seqs;
    if (kept) memcpy(hits, cand, kept * sizeof *cand);
    free(cand);
    if (cnt) cnt->n_prefilter_hits += kept;
    return (int)kept;
}

/* ------------------------------------------------------------------ E5: gapped 3Di+AA Smith-Waterman
 * Foldseek structurealign / SSW: affine local alignment on S3+SA, score + end with
 * the frozen tie-break (smallest tEnd, then smallest qEnd).  rev_q/rev_t read the first lq/lt
 * residues backwards (index l-1-k) — used for the reverse-query pass and the start-position pass. */
void uco_sw(const uint8_t *q3, const uint8_t *qa, int lq, int rev_q,
            const uint8_t *t3, const uint8_t *ta, int lt, int rev_t,
            const uco_params *p, int32_t *score, int32_t *qend, int32_t *tend) {
    const int open = p->gap_open, ext = p->gap_ext;
    /* per-thread scratch (the cpu_baseline leg calls this millions of times from many threads) */
    static __thread int32_t *scratch = NULL;
    static __thread size_t scratch_cap = 0;
    if (scratch_cap < 2 * ((size_t)lq + 1)) {
        free(scratch);
        scratch_cap = 2 * ((size_t)lq + 1) + 256;
        scratch = (int32_t *)malloc(scratch_cap * sizeof(int32_t));
    }
    int32_t *H = scratch;                    /* column j-1, rows -1..lq-1 at [i+1] */
    int32_t *E = scratch + (size_t)lq + 1;
    for (int i = 0; i <= lq; i++) { H[i] = 0; E[i] = -(1 << 28); }
    int32_t best = 0, bq = -1, bt = -1;
    for (int j = 0; j < lt; j++) {
        int tj = rev_t ? lt - 1 - j : j;
        const int8_t *r3 = p->S3 + t3[tj];           /* column of S3 (symmetric use: S[q][t]) */
        const int8_t *ra = p->SA + ta[tj];
        int32_t hdiag = 0, f = -(1 << 28), hup = 0;
        int32_t colbest = 0, colrow = -1;
        for (int i = 0; i < lq; i++) {
            int qi = rev_q ? lq - 1 - i : i;
            int s = r3[q3[qi] * UCO_A] + ra[qa[qi] * UCO_A];
            int32_t hleft = H[i + 1];
            int32_t e = E[i + 1] - ext; if (hleft - open > e) e = hleft - open;
            f = f - ext; if (hup - open > f) f = hup - open;
            int32_t h = hdiag + s;
            if (e > h) h = e;
            if (f > h) h = f;
            if (h < 0) h = 0;
            hdiag = hleft;
            H[i + 1] = h; E[i + 1] = e; hup = h;
            if (h > colbest) { colbest = h; colrow = i; }
        }
        if (colbest > best) { best = colbest; bq = colrow; bt = j; }
    }
    *score = best; *qend = bq; *tend = bt;
}

int uco_can_be_covered(const uco_params *p, int lq, int lt) {
    if (!p->len_gate || !(p->cov > 0.0f)) return 1;
    if (lq <= 0 || lt <= 0) return 0;
    const float ql = (float)lq, tl = (float)lt;
    return p->cov_mode == 0 ? (ql / tl >= p->cov && tl / ql >= p->cov) : p->cov_mode == 1 ? (ql / tl >= p->cov) : (tl / ql >= p->cov);
}

int32_t uco_min_score_q(const uco_params *p, uint32_t q, int lq, uint64_t db_residues) {
    return p->min_score_table ? p->min_score_table[q] : uco_min_score(p, lq, db_residues);
}
int32_t uco_min_score(const uco_params *p, int lq, uint64_t db_residues) {
    double scale = p->K * (double)lq * (double)db_residues;
    int32_t s = 1;
    while (scale * exp(-p->lambda * (double)s) > p->evalue && s < (1 << 30)) s++;
    return s;
}

/* traceback on the [qs..qe]x[ts..te] box (spec E6): returns aln_len and idents */
static void traceback(const uco_db *db, uint32_t q, uint32_t t, const uco_params *p,
                      int qs, int qe, int ts, int te, int32_t *aln_len, int32_t *idents, int32_t *gap_opens) {
    const uint8_t *q3 = db->s3 + db->off[q] + qs, *qa = db->sa + db->off[q] + qs;
    const uint8_t *t3 = db->s3 + db->off[t] + ts, *ta = db->sa + db->off[t] + ts;
    int lq = qe - qs + 1, lt = te - ts + 1;
    const int open = p->gap_open, ext = p->gap_ext;
    const int32_t NEG = -(1 << 28);
    size_t W = (size_t)lt + 1;
    int32_t *H = (int32_t *)calloc(((size_t)lq + 1) * W, sizeof(int32_t));
    int32_t *E = (int32_t *)malloc(((size_t)lq + 1) * W * sizeof(int32_t));
    int32_t *F = (int32_t *)malloc(((size_t)lq + 1) * W * sizeof(int32_t));
    for (size_t k = 0; k < ((size_t)lq + 1) * W; k++) { E[k] = NEG; F[k] = NEG; }
    for (int i = 1; i <= lq; i++)
        for (int j = 1; j <= lt; j++) {
            int s = p->S3[q3[i - 1] * UCO_A + t3[j - 1]] + p->SA[qa[i - 1] * UCO_A + ta[j - 1]];
            int32_t e = E[i * W + j - 1] - ext; if (H[i * W + j - 1] - open > e) e = H[i * W + j - 1] - open;
            int32_t f = F[(i - 1) * W + j] - ext; if (H[(i - 1) * W + j] - open > f) f = H[(i - 1) * W + j] - open;
            int32_t h = H[(i - 1) * W + j - 1] + s;
            if (e > h) h = e;
            if (f > h) h = f;
            if (h < 0) h = 0;
            H[i * W + j] = h; E[i * W + j] = e; F[i * W + j] = f;
        }
    int i = lq, j = lt, state = 0, len = 0, id = 0, gaps = 0;
    while (i > 0 && j > 0) {
        if (state == 0) {
            int32_t h = H[i * W + j];
            if (h == 0) break;
            int s = p->S3[q3[i - 1] * UCO_A + t3[j - 1]] + p->SA[qa[i - 1] * UCO_A + ta[j - 1]];
            if (h == H[(i - 1) * W + j - 1] + s) { len++; id += (qa[i - 1] == ta[j - 1]); i--; j--; }
            else if (h == F[i * W + j]) { state = 1; gaps++; }
            else { state = 2; gaps++; }
        } else if (state == 1) { /* gap consuming query residue i */
            len++;
            if (F[i * W + j] == H[(i - 1) * W + j] - open) state = 0;
            i--;
        } else {                 /* gap consuming target residue j */
            len++;
            if (E[i * W + j] == H[i * W + j - 1] - open) state = 0;
            j--;
        }
    }
    free(H); free(E); free(F);
    *aln_len = len; *idents = id; *gap_opens = gaps;
}

void uco_traceback(const uco_db *db, uint32_t q, uint32_t t, const uco_params *p, int qs, int qe, int ts, int te,
                   int32_t *aln_len, int32_t *idents, int32_t *gap_opens) {
    traceback(db, q, t, p, qs, qe, ts, te, aln_len, idents, gap_opens);
}

/* Foldseek structurealign for one (query, target) pair (call site cluster.rs:45-49). */
void uco_align_pair(const uco_db *db, uint32_t q, uint32_t t, const uco_params *p, int32_t min_score, uco_aln *o) {
    memset(o, 0, sizeof *o);
    const uint8_t *q3 = db->s3 + db->off[q], *qa = db->sa + db->off[q];
    const uint8_t *t3 = db->s3 + db->off[t], *ta = db->sa + db->off[t];
    int lq = (int)(db->off[q + 1] - db->off[q]), lt = (int)(db->off[t + 1] - db->off[t]);
    int32_t qe, te, dq, dt;
    if (!uco_can_be_covered(p, lq, lt)) return;      /* rule UC-1/L (optional): not aligned, all-zero record */
    uco_sw(q3, qa, lq, 0, t3, ta, lt, 0, p, &o->score, &qe, &te);
    /* UC-1.1: corrected <= score, so a pair whose forward score is below the threshold cannot pass and the
       reversed-query pass is not run for it (score_rev stays 0) */
    if (p->rev_correction && o->score >= min_score) uco_sw(q3, qa, lq, 1, t3, ta, lt, 0, p, &o->score_rev, &dq, &dt);
    o->corrected = o->score - o->score_rev;
    o->qend = qe; o->tend = te; o->qstart = -1; o->tstart = -1;
    o->pass_evalue = (o->score > 0 && o->corrected >= min_score);
    if (!o->pass_evalue) return;
    int32_t s2, qe2, te2;
    uco_sw(q3, qa, qe + 1, 1, t3, ta, te + 1, 1, p, &s2, &qe2, &te2);
    o->qstart = qe - qe2; o->tstart = te - te2;
    float qcov = (float)(o->qend - o->qstart + 1) / (float)lq;
    float tcov = (float)(o->tend - o->tstart + 1) / (float)lt;
    int ok = p->cov_mode == 0 ? (qcov >= p->cov && tcov >= p->cov) : p->cov_mode == 1 ? (tcov >= p->cov) : (qcov >= p->cov);
    if (ok && (p->min_seq_id > 0.0f || p->want_tb)) {
        traceback(db, q, t, p, o->qstart, o->qend, o->tstart, o->tend, &o->aln_len, &o->idents, &o->gap_opens);
        float sid = o->aln_len > 0 ? (float)o->idents / (float)o->aln_len : 0.0f;
        if (p->min_seq_id > 0.0f) ok = sid >= p->min_seq_id;
    }
    o->accepted = ok;
}

/* ------------------------------------------------------------------ E7: greedy set cover (MMseqs2 clust, cluster-mode 0) */
typedef struct { uint32_t cnt, id; } hent;
static int hless(hent a, hent b) { return a.cnt != b.cnt ? a.cnt > b.cnt : a.id < b.id; } /* a before b */
typedef struct { hent *h; size_t n, cap; } heap_t;
static void hpush(heap_t *hp, hent x) {
    if (hp->n == hp->cap) { hp->cap = hp->cap ? hp->cap * 2 : 1024; hp->h = (hent *)realloc(hp->h, hp->cap * sizeof(hent)); }
    size_t i = hp->n++;
    while (i > 0) { size_t pa = (i - 1) / 2; if (hless(x, hp->h[pa])) { hp->h[i] = hp->h[pa]; i = pa; } else break; }
    hp->h[i] = x;
}
static hent hpop(heap_t *hp) {
    hent top = hp->h[0], x = hp->h[--hp->n];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i; hent best = x;
        if (l < hp->n && hless(hp->h[l], best)) { m = l; best = hp->h[l]; }
        if (r < hp->n && hless(hp->h[r], best)) { m = r; best = hp->h[r]; }
        if (m == i) break;
        hp->h[i] = hp->h[m]; i = m;
    }
    if (hp->n) hp->h[i] = x;
    return top;
}

int uco_setcover(uint32_t n, const uint32_t *edges, uint64_t n_edges, uint32_t *assign) {
    /* symmetric, de-duplicated adjacency without self loops */
    uint64_t *keys = (uint64_t *)malloc((2 * n_edges + 1) * sizeof(uint64_t));
    uint64_t nk = 0;
    for (uint64_t e = 0; e < n_edges; e++) {
        uint32_t a = edges[2 * e], b = edges[2 * e + 1];
        if (a == b || a >= n || b >= n) continue;
        keys[nk++] = ((uint64_t)a << 32) | b;
        keys[nk++] = ((uint64_t)b << 32) | a;
    }
    if (nk) qsort(keys, nk, sizeof(uint64_t), cmp_u64);
    uint64_t u = 0;
    for (uint64_t k = 0; k < nk; k++) if (k == 0 || keys[k] != keys[k - 1]) keys[u++] = keys[k];
    nk = u;
    uint64_t *aoff = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    for (uint64_t k = 0; k < nk; k++) aoff[(keys[k] >> 32) + 1]++;
    for (uint32_t i = 0; i < n; i++) aoff[i + 1] += aoff[i];
    uint32_t *cnt = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    heap_t hp = {0};
    uint32_t *newly = NULL; size_t newly_cap = 0;
    for (uint32_t i = 0; i < n; i++) { cnt[i] = (uint32_t)(aoff[i + 1] - aoff[i]) + 1; assign[i] = UINT32_MAX; hpush(&hp, (hent){cnt[i], i}); }
    while (hp.n) {
        hent top = hpop(&hp);
        if (assign[top.id] != UINT32_MAX || top.cnt != cnt[top.id]) continue; /* stale */
        uint32_t rep = top.id;
        /* members: rep + its still-unassigned neighbours */
        size_t nn = 0;
        if (newly_cap < (size_t)(aoff[rep + 1] - aoff[rep]) + 1) {
            newly_cap = (size_t)(aoff[rep + 1] - aoff[rep]) + 1;
            newly = (uint32_t *)realloc(newly, newly_cap * sizeof(uint32_t));
        }
        assign[rep] = rep; newly[nn++] = rep;
        for (uint64_t k = aoff[rep]; k < aoff[rep + 1]; k++) {
            uint32_t v = (uint32_t)keys[k];
            if (assign[v] == UINT32_MAX) { assign[v] = rep; newly[nn++] = v; }
        }
        /* every newly covered element leaves the sets of its still-unassigned neighbours */
        for (size_t a = 0; a < nn; a++) {
            uint32_t v = newly[a];
            for (uint64_t m = aoff[v]; m < aoff[v + 1]; m++) {
                uint32_t w = (uint32_t)keys[m];
                if (assign[w] == UINT32_MAX) { cnt[w]--; hpush(&hp, (hent){cnt[w], w}); }
            }
        }
    }
    free(newly); free(hp.h); free(cnt); free(aoff); free(keys);
    return 0;
}

/* ------------------------------------------------------------------ full pipeline == `foldseek cluster` (cluster.rs:45-56) */
int uco_cluster(const uco_db *db, const uco_params *p, int threads, uint32_t *assign, uco_counts *cnt,
                uco_hit *hits_out, uint32_t *hit_cnt_out, uco_aln *aln_out) {
    uco_index ix;
    if (uco_index_build(db, 0, db->n, p, &ix) != 0) return -1;
    const uint32_t n = db->n; const int M = p->max_seqs;
    uco_hit *hits = hits_out ? hits_out : (uco_hit *)malloc((size_t)n * M * sizeof(uco_hit));
    uint32_t *hcnt = hit_cnt_out ? hit_cnt_out : (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uco_counts total; memset(&total, 0, sizeof total);
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        uco_counts loc; memset(&loc, 0, sizeof loc);
#pragma omp for schedule(dynamic, 8)
        for (int64_t q = 0; q < (int64_t)n; q++)
            hcnt[q] = (uint32_t)uco_prefilter_query(db, &ix, (uint32_t)q, p, hits + (size_t)q * M, &loc);
#pragma omp critical
        {
            total.n_sim_kmers += loc.n_sim_kmers; total.n_kmer_hits += loc.n_kmer_hits;
            total.n_candidates += loc.n_candidates; total.n_prefilter_hits += loc.n_prefilter_hits;
        }
    }
    uco_index_free(&ix);
    uint64_t npairs = 0;
    uint64_t *poff = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    for (uint32_t q = 0; q < n; q++) { poff[q] = npairs; npairs += hcnt[q]; }
    poff[n] = npairs;
    uint32_t *edges = (uint32_t *)malloc((2 * npairs + 2) * sizeof(uint32_t));
    uint8_t *acc = (uint8_t *)calloc(npairs + 1, 1);
    const uint64_t dbres = db->off[n];
    uint64_t c_f = 0, c_r = 0, c_s = 0, n_gated = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : c_f, c_r, c_s, n_gated)
    for (int64_t q = 0; q < (int64_t)n; q++) {
        int lq = (int)(db->off[q + 1] - db->off[q]);
        int32_t ms = uco_min_score_q(p, (uint32_t)q, lq, dbres);
        for (uint32_t k = 0; k < hcnt[q]; k++) {
            uco_aln a;
            uint32_t t = hits[(size_t)q * M + k].t;
            uco_align_pair(db, (uint32_t)q, t, p, ms, &a);
            int lt = (int)(db->off[t + 1] - db->off[t]);
            if (!uco_can_be_covered(p, lq, lt)) { n_gated++; if (aln_out) aln_out[(size_t)q * M + k] = a; continue; }
            c_f += (uint64_t)lq * lt; if (p->rev_correction && a.score >= ms) c_r += (uint64_t)lq * lt;
            if (a.pass_evalue) c_s += (uint64_t)(a.qend + 1) * (a.tend + 1);
            acc[poff[q] + k] = (uint8_t)a.accepted;
            if (aln_out) aln_out[(size_t)q * M + k] = a;
        }
    }
    uint64_t ne = 0;
    for (uint32_t q = 0; q < n; q++)
        for (uint32_t k = 0; k < hcnt[q]; k++)
            if (acc[poff[q] + k]) { edges[2 * ne] = q; edges[2 * ne + 1] = hits[(size_t)q * M + k].t; ne++; }
    uco_setcover(n, edges, ne, assign);
    uint64_t ncl = 0;
    for (uint32_t i = 0; i < n; i++) ncl += (assign[i] == i);
    total.n_alignments = npairs - n_gated; total.n_edges = ne; total.n_clusters = ncl;
    total.cells_fwd = c_f; total.cells_rev = c_r; total.cells_start = c_s;
    if (cnt) *cnt = total;
    free(edges); free(acc); free(poff);
    if (!hits_out) free(hits);
    if (!hit_cnt_out) free(hcnt);
    return 0;
}

/* ------------------------------------------------------------------ E9: `foldseek createtsv` (cluster.rs:59-64) */
/* ------------------------------------------------------------------ search path == `foldseek search` + `convertalis` (search.rs:44-61) */
int uco_search(const uco_db *qdb, const uco_db *tdb, const uco_params *p_in, int threads,
               uco_hit *hits_out, uint32_t *hit_cnt_out, uco_aln *aln_out, uco_counts *cnt) {
    uco_params pv = *p_in; pv.want_tb = 1;
    const uco_params *p = &pv;
    const uint32_t nt = tdb->n, nq = qdb->n, n = nt + nq;
    const int M = p->max_seqs;
    /* combined database: targets first, then queries; only [0, nt) is indexed */
    uco_db db; memset(&db, 0, sizeof db);
    db.n = n;
    db.off = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    const uint64_t rt = tdb->off[nt], rq = qdb->off[nq];
    db.s3 = (uint8_t *)malloc(rt + rq + 1); db.sa = (uint8_t *)malloc(rt + rq + 1);
    memcpy(db.s3, tdb->s3, rt); memcpy(db.sa, tdb->sa, rt);
    memcpy(db.s3 + rt, qdb->s3, rq); memcpy(db.sa + rt, qdb->sa, rq);
    for (uint32_t i = 0; i <= nt; i++) db.off[i] = tdb->off[i];
    for (uint32_t i = 0; i <= nq; i++) db.off[nt + i] = rt + qdb->off[i];
    uco_index ix;
    if (uco_index_build(&db, 0, nt, p, &ix) != 0) { free(db.off); free(db.s3); free(db.sa); return -1; }
    uco_counts total; memset(&total, 0, sizeof total);
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        uco_counts loc; memset(&loc, 0, sizeof loc);
#pragma omp for schedule(dynamic, 8)
        for (int64_t q = 0; q < (int64_t)nq; q++)
            hit_cnt_out[q] = (uint32_t)uco_prefilter_query(&db, &ix, nt + (uint32_t)q, p, hits_out + (size_t)q * M, &loc);
#pragma omp critical
        {
            total.n_sim_kmers += loc.n_sim_kmers; total.n_kmer_hits += loc.n_kmer_hits;
            total.n_candidates += loc.n_candidates; total.n_prefilter_hits += loc.n_prefilter_hits;
        }
    }
    uco_index_free(&ix);
    uint64_t npairs = 0, ne = 0, c_f = 0, c_r = 0, c_s = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : c_f, c_r, c_s, npairs, ne)
    for (int64_t q = 0; q < (int64_t)nq; q++) {
        const int lq = (int)(qdb->off[q + 1] - qdb->off[q]);
        const int32_t ms = uco_min_score(p, lq, rt);
        for (uint32_t k = 0; k < hit_cnt_out[q]; k++) {
            uco_aln a;
            const uint32_t t = hits_out[(size_t)q * M + k].t;
            uco_align_pair(&db, nt + (uint32_t)q, t, p, ms, &a);
            const int lt = (int)(tdb->off[t + 1] - tdb->off[t]);
            if (!uco_can_be_covered(p, lq, lt)) { aln_out[(size_t)q * M + k] = a; continue; }
            c_f += (uint64_t)lq * lt; if (p->rev_correction && a.score >= ms) c_r += (uint64_t)lq * lt;
            if (a.pass_evalue) c_s += (uint64_t)(a.qend + 1) * (a.tend + 1);
            npairs++; ne += a.accepted;
            aln_out[(size_t)q * M + k] = a;
        }
    }
    total.n_alignments = npairs; total.n_edges = ne;
    total.cells_fwd = c_f; total.cells_rev = c_r; total.cells_start = c_s;
    if (cnt) *cnt = total;
    free(db.off); free(db.s3); free(db.sa);
    return 0;
}

typedef struct { int32_t corrected; uint32_t t; uint32_t k; } m8ord;
static int m8cmp(const void *a, const void *b) {
    const m8ord *x = (const m8ord *)a, *y = (const m8ord *)b;
    if (x->corrected != y->corrected) return x->corrected > y->corrected ? -1 : 1;
    return x->t < y->t ? -1 : (x->t > y->t ? 1 : 0);
}

int uco_write_m8(const char *path, const uco_db *qdb, const uco_db *tdb, const uco_params *p,
                 const uco_hit *hits, const uint32_t *hit_cnt, const uco_aln *aln) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    const int M = p->max_seqs;
    const double rt = (double)tdb->off[tdb->n];
    m8ord *ord = (m8ord *)malloc((size_t)(M ? M : 1) * sizeof(m8ord));
    for (uint32_t q = 0; q < qdb->n; q++) {
        uint32_t m = 0;
        for (uint32_t k = 0; k < hit_cnt[q]; k++)
            if (aln[(size_t)q * M + k].accepted) { ord[m].corrected = aln[(size_t)q * M + k].corrected; ord[m].t = hits[(size_t)q * M + k].t; ord[m].k = k; m++; }
        if (m) qsort(ord, m, sizeof(m8ord), m8cmp);
        const double lq = (double)(qdb->off[q + 1] - qdb->off[q]);
        for (uint32_t j = 0; j < m; j++) {
            const uco_aln *a = &aln[(size_t)q * M + ord[j].k];
            const int pairs = (a->qend - a->qstart + 1) + (a->tend - a->tstart + 1) - a->aln_len;
            const double fident = a->aln_len > 0 ? (double)a->idents / (double)a->aln_len : 0.0;
            const double ev = p->K * lq * rt * exp(-p->lambda * (double)a->corrected);
            const int bits = (int)((p->lambda * (double)a->corrected - log(p->K)) / log(2.0));
            if (qdb->names) fprintf(f, "%s\t", qdb->names[q]); else fprintf(f, "%u\t", q);
            if (tdb->names) fprintf(f, "%s\t", tdb->names[ord[j].t]); else fprintf(f, "%u\t", ord[j].t);
            fprintf(f, "%.3f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%.3E\t%d\n", fident, a->aln_len, pairs - a->idents, a->gap_opens,
                    a->qstart + 1, a->qend + 1, a->tstart + 1, a->tend + 1, ev, bits);
        }
    }
    free(ord);
    return fclose(f) == 0 ? 0 : -1;
}

/* ------------------------------------------------------------------ E8a: linear-time pre-step (Linclust restated) */
static inline uint64_t lc_hash(uint32_t v) {
    uint64_t z = (uint64_t)v + 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}
typedef struct { uint64_t h; uint32_t v; uint32_t pos; } lc_cand;
typedef struct { uint32_t v, seq; } lc_ent;
static int lc_cand_cmp(const void *a, const void *b) {
    const lc_cand *x = (const lc_cand *)a, *y = (const lc_cand *)b;
    if (x->h != y->h) return x->h < y->h ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}
static int lc_ent_cmp(const void *a, const void *b) {
    const lc_ent *x = (const lc_ent *)a, *y = (const lc_ent *)b;
    if (x->v != y->v) return x->v < y->v ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}
static int lc_pair_cmp(const void *a, const void *b) {
    const uint32_t *x = (const uint32_t *)a, *y = (const uint32_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return x[1] < y[1] ? -1 : (x[1] > y[1] ? 1 : 0);
}

int uco_linclust_pairs(const uco_db *db, const uco_params *p, int m, uint32_t **pairs_out, uint64_t *n_pairs) {
    int off[UCO_K];
    const int span = uco_pattern_offsets(p->pattern, off);
    if (span < 0 || m < 1) return -1;
    const uint32_t n = db->n;
    lc_ent *ent = (lc_ent *)malloc(((size_t)n * m + 1) * sizeof(lc_ent));
    size_t ne = 0, ccap = 1024;
    lc_cand *cand = (lc_cand *)malloc(ccap * sizeof(lc_cand));
    for (uint32_t t = 0; t < n; t++) {
        const uint8_t *s = db->s3 + db->off[t];
        const int64_t l = (int64_t)(db->off[t + 1] - db->off[t]);
        size_t nc = 0;
        for (int64_t j = 0; j + span <= l && j <= 65535; j++) {
            uint32_t v;
            if (!kmer_at(s + j, off, &v)) continue;
            if (nc == ccap) { ccap *= 2; cand = (lc_cand *)realloc(cand, ccap * sizeof(lc_cand)); }
            cand[nc].h = lc_hash(v); cand[nc].v = v; cand[nc].pos = (uint32_t)j; nc++;
        }
        if (nc) qsort(cand, nc, sizeof(lc_cand), lc_cand_cmp);
        for (size_t k = 0; k < nc && k < (size_t)m; k++) { ent[ne].v = cand[k].v; ent[ne].seq = t; ne++; }
    }
    free(cand);
    if (ne) qsort(ent, ne, sizeof(lc_ent), lc_ent_cmp);
    uint32_t *pairs = (uint32_t *)malloc((2 * ne + 2) * sizeof(uint32_t));
    uint64_t np = 0;
    for (size_t b = 0; b < ne;) {
        size_t e = b;
        while (e < ne && ent[e].v == ent[b].v) e++;
        /* centre: longest sequence of the group, ties: smallest id (entries are sorted by id) */
        uint32_t c = ent[b].seq;
        uint64_t lc = db->off[c + 1] - db->off[c];
        for (size_t k = b + 1; k < e; k++) {
            const uint32_t sq = ent[k].seq;
            const uint64_t ls = db->off[sq + 1] - db->off[sq];
            if (ls > lc) { c = sq; lc = ls; }
        }
        for (size_t k = b; k < e; k++)
            if (ent[k].seq != c && (k == b || ent[k].seq != ent[k - 1].seq)) { pairs[2 * np] = c; pairs[2 * np + 1] = ent[k].seq; np++; }
        b = e;
    }
    free(ent);
    if (np) qsort(pairs, np, 2 * sizeof(uint32_t), lc_pair_cmp);
    uint64_t w = 0;
    for (uint64_t k = 0; k < np; k++)
        if (k == 0 || pairs[2 * k] != pairs[2 * k - 2] || pairs[2 * k + 1] != pairs[2 * k - 1]) { pairs[2 * w] = pairs[2 * k]; pairs[2 * w + 1] = pairs[2 * k + 1]; w++; }
    if (n_pairs) *n_pairs = w;
    if (pairs_out) *pairs_out = pairs; else free(pairs);
    return 0;
}

int uco_cluster_linclust(const uco_db *db, const uco_params *p, int m, int threads, uint32_t *assign, uco_counts *cnt) {
    uint32_t *pairs = NULL;
    uint64_t np = 0;
    if (uco_linclust_pairs(db, p, m, &pairs, &np) != 0) return -1;
    const uint32_t n = db->n;
    const uint64_t dbres = db->off[n];
    uint8_t *acc = (uint8_t *)calloc(np + 1, 1);
    uint64_t c_f = 0, c_r = 0, c_s = 0, n_gated = 0;
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : c_f, c_r, c_s, n_gated)
    for (int64_t k = 0; k < (int64_t)np; k++) {
        const uint32_t q = pairs[2 * k], t = pairs[2 * k + 1];
        const int lq = (int)(db->off[q + 1] - db->off[q]), lt = (int)(db->off[t + 1] - db->off[t]);
        const int32_t ms = uco_min_score(p, lq, dbres);
        uco_aln a;
        uco_align_pair(db, q, t, p, ms, &a);
        if (!uco_can_be_covered(p, lq, lt)) { n_gated++; continue; }
        c_f += (uint64_t)lq * lt; if (p->rev_correction && a.score >= ms) c_r += (uint64_t)lq * lt;
        if (a.pass_evalue) c_s += (uint64_t)(a.qend + 1) * (a.tend + 1);
        acc[k] = (uint8_t)a.accepted;
    }
    uint32_t *edges = (uint32_t *)malloc((2 * np + 2) * sizeof(uint32_t));
    uint64_t ne = 0;
    for (uint64_t k = 0; k < np; k++)
        if (acc[k]) { edges[2 * ne] = pairs[2 * k]; edges[2 * ne + 1] = pairs[2 * k + 1]; ne++; }
    uco_setcover(n, edges, ne, assign);
    if (cnt) {
        memset(cnt, 0, sizeof *cnt);
        uint64_t ncl = 0;
        for (uint32_t i = 0; i < n; i++) ncl += assign[i] == i;
        cnt->n_prefilter_hits = np; cnt->n_alignments = np - n_gated; cnt->n_edges = ne; cnt->n_clusters = ncl;
        cnt->cells_fwd = c_f; cnt->cells_rev = c_r; cnt->cells_start = c_s;
    }
    free(edges); free(acc); free(pairs);
    return 0;
}

static void uco_subdb(const uco_db *db, const uint32_t *ids, uint32_t k, uco_db *sub) {
    memset(sub, 0, sizeof *sub);
    sub->n = k;
    sub->off = (uint64_t *)malloc(((size_t)k + 1) * sizeof(uint64_t));
    uint64_t tot = 0;
    for (uint32_t i = 0; i < k; i++) { sub->off[i] = tot; tot += db->off[ids[i] + 1] - db->off[ids[i]]; }
    sub->off[k] = tot;
    sub->s3 = (uint8_t *)malloc(tot ? tot : 1); sub->sa = (uint8_t *)malloc(tot ? tot : 1);
    for (uint32_t i = 0; i < k; i++) {
        const uint64_t len = db->off[ids[i] + 1] - db->off[ids[i]];
        memcpy(sub->s3 + sub->off[i], db->s3 + db->off[ids[i]], len);
        memcpy(sub->sa + sub->off[i], db->sa + db->off[ids[i]], len);
    }
}

int uco_cluster_workflow(const uco_db *db, const uco_params *p, int linclust_m, int steps, const int *thr, int threads,
                         uint32_t *assign, uco_counts *cnt, uint32_t *round_sizes) {
    const uint32_t n = db->n;
    if (steps < 0 || (steps == 0 && linclust_m <= 0)) return -1;
    uint32_t *cur = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t *posmap = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t ncur = n;
    for (uint32_t i = 0; i < n; i++) { cur[i] = i; assign[i] = i; }
    uco_counts total; memset(&total, 0, sizeof total);
    int rc = 0;
    const int rounds = steps + (linclust_m > 0 ? 1 : 0);
    for (int r = 0; r < rounds && rc == 0; r++) {
        if (round_sizes) round_sizes[r] = ncur;
        uco_db sub;
        uco_subdb(db, cur, ncur, &sub);
        uint32_t *sa_ = (uint32_t *)malloc((size_t)(ncur ? ncur : 1) * sizeof(uint32_t));
        uco_counts c; memset(&c, 0, sizeof c);
        if (linclust_m > 0 && r == 0) rc = uco_cluster_linclust(&sub, p, linclust_m, threads, sa_, &c);
        else {
            uco_params pr = *p;
            pr.kmer_thr = thr[r - (linclust_m > 0 ? 1 : 0)];
            rc = uco_cluster(&sub, &pr, threads, sa_, &c, NULL, NULL, NULL);
        }
        if (rc == 0) {
            total.n_sim_kmers += c.n_sim_kmers; total.n_kmer_hits += c.n_kmer_hits; total.n_candidates += c.n_candidates;
            total.n_prefilter_hits += c.n_prefilter_hits; total.n_alignments += c.n_alignments; total.n_edges += c.n_edges;
            total.cells_fwd += c.cells_fwd; total.cells_rev += c.cells_rev; total.cells_start += c.cells_start;
            for (uint32_t i = 0; i < ncur; i++) posmap[cur[i]] = i;
            for (uint32_t x = 0; x < n; x++) assign[x] = cur[sa_[posmap[assign[x]]]];
            uint32_t k = 0;
            for (uint32_t i = 0; i < ncur; i++) if (sa_[i] == i) cur[k++] = cur[i];
            ncur = k;
        }
        free(sa_); free(sub.off); free(sub.s3); free(sub.sa);
    }
    total.n_clusters = ncur;
    if (cnt) *cnt = total;
    free(cur); free(posmap);
    return rc;
}

/* ------------------------------------------------------------------ E8: cascade (rounds on representatives + merge) */
int uco_cluster_cascade(const uco_db *db, const uco_params *p, int steps, const int *thr, int threads,
                        uint32_t *assign, uco_counts *cnt, uint32_t *round_sizes) {
    const uint32_t n = db->n;
    if (steps < 1) return -1;
    uint32_t *cur = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t *posmap = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t ncur = n;
    for (uint32_t i = 0; i < n; i++) { cur[i] = i; assign[i] = i; }
    uco_counts total; memset(&total, 0, sizeof total);
    int rc = 0;
    for (int r = 0; r < steps && rc == 0; r++) {
        if (round_sizes) round_sizes[r] = ncur;
        uco_db sub; memset(&sub, 0, sizeof sub);
        sub.n = ncur;
        sub.off = (uint64_t *)malloc(((size_t)ncur + 1) * sizeof(uint64_t));
        uint64_t tot = 0;
        for (uint32_t i = 0; i < ncur; i++) { sub.off[i] = tot; tot += db->off[cur[i] + 1] - db->off[cur[i]]; }
        sub.off[ncur] = tot;
        sub.s3 = (uint8_t *)malloc(tot ? tot : 1); sub.sa = (uint8_t *)malloc(tot ? tot : 1);
        for (uint32_t i = 0; i < ncur; i++) {
            const uint64_t len = db->off[cur[i] + 1] - db->off[cur[i]];
            memcpy(sub.s3 + sub.off[i], db->s3 + db->off[cur[i]], len);
            memcpy(sub.sa + sub.off[i], db->sa + db->off[cur[i]], len);
        }
        uco_params pr = *p;
        pr.kmer_thr = thr[r];
        uint32_t *sa_ = (uint32_t *)malloc((size_t)(ncur ? ncur : 1) * sizeof(uint32_t));
        uco_counts c; memset(&c, 0, sizeof c);
        rc = uco_cluster(&sub, &pr, threads, sa_, &c, NULL, NULL, NULL);
        if (rc == 0) {
            total.n_sim_kmers += c.n_sim_kmers; total.n_kmer_hits += c.n_kmer_hits; total.n_candidates += c.n_candidates;
            total.n_prefilter_hits += c.n_prefilter_hits; total.n_alignments += c.n_alignments; total.n_edges += c.n_edges;
            total.cells_fwd += c.cells_fwd; total.cells_rev += c.cells_rev; total.cells_start += c.cells_start;
            for (uint32_t i = 0; i < ncur; i++) posmap[cur[i]] = i;
            for (uint32_t x = 0; x < n; x++) assign[x] = cur[sa_[posmap[assign[x]]]];
            uint32_t k = 0;
            for (uint32_t i = 0; i < ncur; i++) if (sa_[i] == i) cur[k++] = cur[i];
            ncur = k;
        }
        free(sa_); free(sub.off); free(sub.s3); free(sub.sa);
    }
    total.n_clusters = ncur;
    if (cnt) *cnt = total;
    free(cur); free(posmap);
    return rc;
}

int uco_write_tsv(const char *path, const uco_db *db, const uint32_t *assign) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    const uint32_t n = db->n;
    /* members of each representative in ascending id: counting sort by rep */
    uint64_t *off = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    for (uint32_t i = 0; i < n; i++) off[assign[i] + 1]++;
    for (uint32_t i = 0; i < n; i++) off[i + 1] += off[i];
    uint32_t *mem = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uint64_t *cur = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    memcpy(cur, off, ((size_t)n + 1) * sizeof(uint64_t));
    for (uint32_t i = 0; i < n; i++) mem[cur[assign[i]]++] = i;
    for (uint32_t r = 0; r < n; r++) {
        if (off[r + 1] == off[r]) continue;
        fprintf(f, "%s\t%s\n", db->names[r], db->names[r]);
        for (uint64_t k = off[r]; k < off[r + 1]; k++)
            if (mem[k] != r) fprintf(f, "%s\t%s\n", db->names[r], db->names[mem[k]]);
    }
    free(off); free(mem); free(cur);
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ bench.py cpu_baseline leg */
#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

uint64_t uco_sample_run(const uco_db *db, const uco_index *ix, const uco_params *p, int threads,
                        const uint32_t *queries, uint32_t n_queries, double seconds[2]) {
    const int M = p->max_seqs;
    uco_hit *hits = (uco_hit *)malloc((size_t)n_queries * M * sizeof(uco_hit));
    uint32_t *hcnt = (uint32_t *)calloc(n_queries, sizeof(uint32_t));
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t k = 0; k < (int64_t)n_queries; k++)
        hcnt[k] = (uint32_t)uco_prefilter_query(db, ix, queries[k], p, hits + (size_t)k * M, NULL);
    double t1 = now_s();
    const uint64_t dbres = db->off[db->n];
    /* flatten (query, hit) pairs: per-query work is heavy-tailed (long queries x long hit lists), so the
       parallel loop runs over pairs, not queries */
    uint64_t pairs = 0, acc = 0;
    uint64_t *poff = (uint64_t *)malloc(((size_t)n_queries + 1) * sizeof(uint64_t));
    for (uint32_t k = 0; k < n_queries; k++) { poff[k] = pairs; pairs += hcnt[k]; }
    poff[n_queries] = pairs;
    uint32_t *pq = (uint32_t *)malloc((pairs + 1) * sizeof(uint32_t)), *pt = (uint32_t *)malloc((pairs + 1) * sizeof(uint32_t));
    int32_t *pms = (int32_t *)malloc((pairs + 1) * sizeof(int32_t));
    for (uint32_t k = 0; k < n_queries; k++) {
        const uint32_t q = queries[k];
        const int32_t ms = uco_min_score_q(p, q, (int)(db->off[q + 1] - db->off[q]), dbres);
        for (uint32_t h = 0; h < hcnt[k]; h++) { pq[poff[k] + h] = q; pt[poff[k] + h] = hits[(size_t)k * M + h].t; pms[poff[k] + h] = ms; }
    }
    uint64_t gated = 0;   /* rule UC-1/L (optional): pairs the length gate rules out are not alignments */
    for (uint64_t i = 0; i < pairs; i++)
        gated += !uco_can_be_covered(p, (int)(db->off[pq[i] + 1] - db->off[pq[i]]), (int)(db->off[pt[i] + 1] - db->off[pt[i]]));
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : acc)
    for (int64_t i = 0; i < (int64_t)pairs; i++) {
        uco_aln a;
        uco_align_pair(db, pq[i], pt[i], p, pms[i], &a);
        acc += (uint64_t)a.accepted;
    }
    free(poff); free(pq); free(pt); free(pms);
    double t2 = now_s();
    seconds[0] = t1 - t0; seconds[1] = t2 - t1;
    free(hits); free(hcnt);
    return pairs - gated + 0 * acc;
}
