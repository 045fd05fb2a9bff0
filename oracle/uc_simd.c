/* uc_simd.c — vectorised CPU leg for bench.py's `cpu_baseline` ("kind": "simd").  TEST/BENCH INFRASTRUCTURE like the
 * rest of oracle/: nothing in the product links or calls it.
 *
 * Why it exists: the oracle (uc_oracle.c) is a scalar restatement written to be read and checked, not to be fast, so a
 * GPU/oracle ratio says little.  This file runs the SAME spec (UC-1.1: forward pass, reversed-query pass for pairs whose
 * forward score reaches the E-value threshold, start pass + coverage / seq-id gates for those that pass) on the SAME pair
 * list with the gapped stage as inter-sequence SIMD Smith-Waterman in the manner of SWIPE (Rognes 2011, BMC Bioinformatics
 * 12:221): one query against 16 of its targets at a time, one target per int16 lane of an AVX2 register, a per-column
 * score profile over the 21 query letters, affine gaps with the recurrence of uco_sw.  The tie-break (smallest tEnd,
 * then smallest qEnd) is uco_sw's, so every result equals the oracle's — tests/test_oracle_kat.py checks that record by
 * record.  The prefilter (E2-E4: index gathers) is the oracle's scalar code: it is gather-bound, not arithmetic-bound.
 *
 * Compiled with -march=x86-64-v3 (AVX2), never -march=native: the .so is built in the dev container and travels to the
 * GPU box, whose host CPU is a different model. */
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "uc_oracle.h"

#define L 16                      /* int16 lanes */
#define PADSCORE (-200)           /* score of a column beyond the end of a lane's target: nothing extends through it */
#define OVF 32000                 /* best scores at or above this are recomputed by the scalar int32 oracle */

typedef struct {
    const uint8_t *t3, *ta;       /* target letters (forward orientation, first residue) */
    int lt;                       /* columns to run (forward: target length; start pass: tEnd + 1) */
    int rev_t;                    /* read the first lt residues backwards */
    int skip;                     /* start pass: rows (of the reversed full query) before this one are masked */
} lane_t;

typedef struct { int32_t score, qend, tend; } res_t;

/* DP cells of the alignments handed to sw_batch16 (rows x columns of every lane's own problem: what a GCUPS figure divides),
 * and cells the 16-lane batches actually swept (padding to the longest lane included); bench.py reads and resets them */
static uint64_t g_cells_useful, g_cells_swept;
void uco_simd_cells(uint64_t out[2]) {
    out[0] = __atomic_exchange_n(&g_cells_useful, 0, __ATOMIC_RELAXED);
    out[1] = __atomic_exchange_n(&g_cells_swept, 0, __ATOMIC_RELAXED);
}

/* one batch: query (q3,qa,lq; rev_q reads it backwards) against up to 16 lanes.  out[l] = (score, qend, tend) with qend
 * counted from the lane's first unmasked row. */
static void sw_batch16(const uint8_t *q3, const uint8_t *qa, int lq, int rev_q, const lane_t *ln, int nl, const uco_params *p,
                       int16_t *Hbuf, int16_t *Ebuf, res_t *out) {
    int maxlt = 0, r0 = lq;
    for (int l = 0; l < nl; l++) { if (ln[l].lt > maxlt) maxlt = ln[l].lt; if (ln[l].skip < r0) r0 = ln[l].skip; }
    if (nl == 0 || maxlt == 0) { for (int l = 0; l < nl; l++) { out[l].score = 0; out[l].qend = -1; out[l].tend = -1; } return; }
    {
        uint64_t useful = 0;
        for (int l = 0; l < nl; l++) useful += (uint64_t)(lq - ln[l].skip) * (uint64_t)ln[l].lt;
        __atomic_fetch_add(&g_cells_useful, useful, __ATOMIC_RELAXED);
        __atomic_fetch_add(&g_cells_swept, (uint64_t)(lq - r0) * (uint64_t)maxlt * L, __ATOMIC_RELAXED);
    }
    const __m256i vopen = _mm256_set1_epi16((short)p->gap_open), vext = _mm256_set1_epi16((short)p->gap_ext);
    const __m256i vzero = _mm256_setzero_si256(), vneg = _mm256_set1_epi16(-16000);
    int16_t skipv[L];
    for (int l = 0; l < L; l++) skipv[l] = (int16_t)(l < nl ? ln[l].skip : 32767);
    const __m256i vskip = _mm256_loadu_si256((const __m256i *)skipv);
    int any_skip = 0;
    for (int l = 0; l < nl; l++) any_skip |= ln[l].skip != r0;
    __m256i *H = (__m256i *)Hbuf, *E = (__m256i *)Ebuf;
    for (int i = r0; i < lq; i++) { H[i] = vzero; E[i] = vneg; }
    __m256i best = vzero, bq = _mm256_set1_epi16(-1), bt = _mm256_set1_epi16(-1);
    __m256i prof3[UCO_A], profa[UCO_A];
    int16_t tmp3[UCO_A][L], tmpa[UCO_A][L];
    for (int j = 0; j < maxlt; j++) {
        for (int l = 0; l < L; l++) {
            if (l < nl && j < ln[l].lt) {
                const int tj = ln[l].rev_t ? ln[l].lt - 1 - j : j;
                const int8_t *c3 = p->S3 + ln[l].t3[tj], *ca = p->SA + ln[l].ta[tj];
                for (int a = 0; a < UCO_A; a++) { tmp3[a][l] = c3[a * UCO_A]; tmpa[a][l] = ca[a * UCO_A]; }
            } else {
                for (int a = 0; a < UCO_A; a++) { tmp3[a][l] = PADSCORE; tmpa[a][l] = 0; }
            }
        }
        for (int a = 0; a < UCO_A; a++) { prof3[a] = _mm256_loadu_si256((const __m256i *)tmp3[a]); profa[a] = _mm256_loadu_si256((const __m256i *)tmpa[a]); }
        __m256i hdiag = vzero, f = vneg, hup = vzero, colbest = vzero, colrow = _mm256_set1_epi16(-1);
        for (int i = r0; i < lq; i++) {
            const int qi = rev_q ? lq - 1 - i : i;
            const __m256i s = _mm256_add_epi16(prof3[q3[qi]], profa[qa[qi]]);
            const __m256i hleft = H[i];
            const __m256i e = _mm256_max_epi16(_mm256_subs_epi16(E[i], vext), _mm256_subs_epi16(hleft, vopen));
            f = _mm256_max_epi16(_mm256_subs_epi16(f, vext), _mm256_subs_epi16(hup, vopen));
            __m256i h = _mm256_max_epi16(_mm256_max_epi16(_mm256_adds_epi16(hdiag, s), e), _mm256_max_epi16(f, vzero));
            const __m256i vi = _mm256_set1_epi16((short)i);
            if (any_skip) h = _mm256_andnot_si256(_mm256_cmpgt_epi16(vskip, vi), h);   /* rows before a lane's prefix start: H = 0 */
            hdiag = hleft;
            H[i] = h; E[i] = e; hup = h;
            const __m256i m = _mm256_cmpgt_epi16(h, colbest);
            colbest = _mm256_max_epi16(colbest, h);
            colrow = _mm256_blendv_epi8(colrow, vi, m);
        }
        const __m256i m2 = _mm256_cmpgt_epi16(colbest, best);
        best = _mm256_max_epi16(best, colbest);
        bq = _mm256_blendv_epi8(bq, colrow, m2);
        bt = _mm256_blendv_epi8(bt, _mm256_set1_epi16((short)j), m2);
    }
    int16_t sb[L], sq[L], st[L];
    _mm256_storeu_si256((__m256i *)sb, best); _mm256_storeu_si256((__m256i *)sq, bq); _mm256_storeu_si256((__m256i *)st, bt);
    for (int l = 0; l < nl; l++) { out[l].score = sb[l]; out[l].qend = sq[l] < 0 ? -1 : sq[l] - ln[l].skip; out[l].tend = st[l]; }
}

typedef struct { uint32_t idx; int key; } ord_t;
static int ord_cmp(const void *a, const void *b) {
    const ord_t *x = (const ord_t *)a, *y = (const ord_t *)b;
    return x->key != y->key ? (x->key > y->key ? -1 : 1) : (x->idx < y->idx ? -1 : x->idx > y->idx);
}

/* E5/E6 of one query against its hit list (spec UC-1.1), results in out[0..nh) in hit order; identical to uco_align_pair */
void uco_simd_align_query(const uco_db *db, uint32_t q, const uint32_t *targets, uint32_t nh, const uco_params *p, int32_t min_score,
                          uco_aln *out) {
    const uint8_t *q3 = db->s3 + db->off[q], *qa = db->sa + db->off[q];
    const int lq = (int)(db->off[q + 1] - db->off[q]);
    memset(out, 0, (size_t)nh * sizeof(uco_aln));
    if (!nh) return;
    if (lq >= 32000) { for (uint32_t h = 0; h < nh; h++) uco_align_pair(db, q, targets[h], p, min_score, &out[h]); return; }
    int16_t *Hbuf = (int16_t *)aligned_alloc(32, ((size_t)lq + 1) * L * sizeof(int16_t));
    int16_t *Ebuf = (int16_t *)aligned_alloc(32, ((size_t)lq + 1) * L * sizeof(int16_t));
    ord_t *ord = (ord_t *)malloc((size_t)nh * sizeof(ord_t));
    uint8_t *redo = (uint8_t *)calloc(nh, 1);
    lane_t ln[L];
    res_t rs[L];
    /* pass 1: forward, targets grouped by length; rule UC-1/L (optional): pairs the length gate rules out keep their all-zero record */
    uint8_t *gated = (uint8_t *)calloc(nh, 1);
    uint32_t n1 = 0;
    for (uint32_t h = 0; h < nh; h++) {
        const int lt = (int)(db->off[targets[h] + 1] - db->off[targets[h]]);
        if (!uco_can_be_covered(p, lq, lt)) { gated[h] = 1; continue; }
        ord[n1].idx = h; ord[n1].key = lt; n1++;
    }
    if (n1) qsort(ord, n1, sizeof(ord_t), ord_cmp);
    for (uint32_t b = 0; b < n1; b += L) {
        const int nl = (int)(n1 - b < L ? n1 - b : L);
        for (int l = 0; l < nl; l++) {
            const uint32_t t = targets[ord[b + l].idx];
            ln[l].t3 = db->s3 + db->off[t]; ln[l].ta = db->sa + db->off[t]; ln[l].lt = ord[b + l].key; ln[l].rev_t = 0; ln[l].skip = 0;
        }
        sw_batch16(q3, qa, lq, 0, ln, nl, p, Hbuf, Ebuf, rs);
        for (int l = 0; l < nl; l++) {
            uco_aln *o = &out[ord[b + l].idx];
            o->score = rs[l].score; o->qend = rs[l].qend; o->tend = rs[l].tend; o->qstart = -1; o->tstart = -1;
            if (rs[l].score >= OVF || ln[l].lt >= 32000) redo[ord[b + l].idx] = 1;
        }
    }
    /* pass 2: reversed query for the pairs whose forward score reaches the threshold (UC-1.1) */
    uint32_t n2 = 0;
    for (uint32_t h = 0; h < nh; h++)
        if (!gated[h] && !redo[h] && p->rev_correction && out[h].score >= min_score) { ord[n2].idx = h; ord[n2].key = (int)(db->off[targets[h] + 1] - db->off[targets[h]]); n2++; }
    if (n2) qsort(ord, n2, sizeof(ord_t), ord_cmp);
    for (uint32_t b = 0; b < n2; b += L) {
        const int nl = (int)(n2 - b < L ? n2 - b : L);
        for (int l = 0; l < nl; l++) {
            const uint32_t t = targets[ord[b + l].idx];
            ln[l].t3 = db->s3 + db->off[t]; ln[l].ta = db->sa + db->off[t]; ln[l].lt = ord[b + l].key; ln[l].rev_t = 0; ln[l].skip = 0;
        }
        sw_batch16(q3, qa, lq, 1, ln, nl, p, Hbuf, Ebuf, rs);
        for (int l = 0; l < nl; l++) { out[ord[b + l].idx].score_rev = rs[l].score; if (rs[l].score >= OVF) redo[ord[b + l].idx] = 1; }
    }
    for (uint32_t h = 0; h < nh; h++) {
        if (gated[h]) continue;
        out[h].corrected = out[h].score - out[h].score_rev;
        out[h].pass_evalue = out[h].score > 0 && out[h].corrected >= min_score;
    }
    /* pass 3: start positions on the reversed prefixes, grouped by prefix area */
    uint32_t n3 = 0;
    for (uint32_t h = 0; h < nh; h++)
        if (!redo[h] && out[h].pass_evalue) { ord[n3].idx = h; ord[n3].key = out[h].tend + 1; n3++; }
    if (n3) qsort(ord, n3, sizeof(ord_t), ord_cmp);
    for (uint32_t b = 0; b < n3; b += L) {
        const int nl = (int)(n3 - b < L ? n3 - b : L);
        for (int l = 0; l < nl; l++) {
            const uco_aln *o = &out[ord[b + l].idx];
            const uint32_t t = targets[ord[b + l].idx];
            ln[l].t3 = db->s3 + db->off[t]; ln[l].ta = db->sa + db->off[t]; ln[l].lt = o->tend + 1; ln[l].rev_t = 1; ln[l].skip = lq - 1 - o->qend;
        }
        sw_batch16(q3, qa, lq, 1, ln, nl, p, Hbuf, Ebuf, rs);
        for (int l = 0; l < nl; l++) {
            uco_aln *o = &out[ord[b + l].idx];
            o->qstart = o->qend - rs[l].qend; o->tstart = o->tend - rs[l].tend;
        }
    }
    /* gates; the seq-id traceback (few pairs) and anything near the int16 range go through the scalar oracle */
    for (uint32_t h = 0; h < nh; h++) {
        uco_aln *o = &out[h];
        if (gated[h]) continue;
        if (redo[h]) { uco_align_pair(db, q, targets[h], p, min_score, o); continue; }
        if (!o->pass_evalue) continue;
        const int lt = (int)(db->off[targets[h] + 1] - db->off[targets[h]]);
        const float qcov = (float)(o->qend - o->qstart + 1) / (float)lq, tcov = (float)(o->tend - o->tstart + 1) / (float)lt;
        int ok = p->cov_mode == 0 ? (qcov >= p->cov && tcov >= p->cov) : p->cov_mode == 1 ? (tcov >= p->cov) : (qcov >= p->cov);
        if (ok && (p->min_seq_id > 0.0f || p->want_tb)) {
            uco_traceback(db, q, targets[h], p, o->qstart, o->qend, o->tstart, o->tend, &o->aln_len, &o->idents, &o->gap_opens);
            const float sid = o->aln_len > 0 ? (float)o->idents / (float)o->aln_len : 0.0f;
            if (p->min_seq_id > 0.0f) ok = sid >= p->min_seq_id;
        }
        o->accepted = ok;
    }
    free(Hbuf); free(Ebuf); free(ord); free(redo); free(gated);
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

/* the cpu_baseline leg: E2-E6 for the listed queries against a prebuilt index; same contract as uco_sample_run.
 * aln_out (optional, n_queries * max_seqs) / hit_out / cnt_out receive the records for the parity test. */
uint64_t uco_simd_sample_run(const uco_db *db, const uco_index *ix, const uco_params *p, int threads,
                             const uint32_t *queries, uint32_t n_queries, double seconds[2],
                             uco_hit *hit_out, uint32_t *cnt_out, uco_aln *aln_out) {
    return uco_simd_sample_run_counts(db, ix, p, threads, queries, n_queries, seconds, hit_out, cnt_out, aln_out, NULL);
}

/* the same with the prefilter's stage counters of these queries (similar k-mers, k-mer hits, candidates, prefilter hits) added
 * to *pc: tools/oracle_at_size.py sums them over query chunks to get the whole-database counters of uco_cluster */
uint64_t uco_simd_sample_run_counts(const uco_db *db, const uco_index *ix, const uco_params *p, int threads,
                                    const uint32_t *queries, uint32_t n_queries, double seconds[2],
                                    uco_hit *hit_out, uint32_t *cnt_out, uco_aln *aln_out, uco_counts *pc) {
    const int M = p->max_seqs;
    uco_hit *hits = hit_out ? hit_out : (uco_hit *)malloc((size_t)n_queries * M * sizeof(uco_hit));
    uint32_t *hcnt = cnt_out ? cnt_out : (uint32_t *)calloc(n_queries, sizeof(uint32_t));
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    const double t0 = now_s();
#pragma omp parallel
    {
        uco_counts loc; memset(&loc, 0, sizeof loc);
#pragma omp for schedule(dynamic, 4)
        for (int64_t k = 0; k < (int64_t)n_queries; k++)
            hcnt[k] = (uint32_t)uco_prefilter_query(db, ix, queries[k], p, hits + (size_t)k * M, pc ? &loc : NULL);
        if (pc) {
#pragma omp critical
            {
                pc->n_sim_kmers += loc.n_sim_kmers; pc->n_kmer_hits += loc.n_kmer_hits;
                pc->n_candidates += loc.n_candidates; pc->n_prefilter_hits += loc.n_prefilter_hits;
            }
        }
    }
    const double t1 = now_s();
    const uint64_t dbres = db->off[db->n];
    uint64_t pairs = 0;
    for (uint32_t k = 0; k < n_queries; k++) pairs += hcnt[k];
    if (p->len_gate)      /* rule UC-1/L (optional): pairs the length gate rules out are not alignments */
        for (uint32_t k = 0; k < n_queries; k++)
            for (uint32_t h = 0; h < hcnt[k]; h++) {
                const uint32_t q = queries[k], t = hits[(size_t)k * M + h].t;
                pairs -= !uco_can_be_covered(p, (int)(db->off[q + 1] - db->off[q]), (int)(db->off[t + 1] - db->off[t]));
            }
    /* one task = one query with its whole hit list (its targets share the query profile); heavy queries first */
    ord_t *qo = (ord_t *)malloc(((size_t)n_queries + 1) * sizeof(ord_t));
    for (uint32_t k = 0; k < n_queries; k++) { qo[k].idx = k; qo[k].key = (int)((db->off[queries[k] + 1] - db->off[queries[k]]) * (uint64_t)hcnt[k] >> 6); }
    if (n_queries) qsort(qo, n_queries, sizeof(ord_t), ord_cmp);
#pragma omp parallel
    {
        uco_aln *buf = aln_out ? NULL : (uco_aln *)malloc((size_t)M * sizeof(uco_aln));
        uint32_t *tg = (uint32_t *)malloc((size_t)M * sizeof(uint32_t));
#pragma omp for schedule(dynamic, 1)
        for (int64_t kk = 0; kk < (int64_t)n_queries; kk++) {
            const uint32_t k = qo[kk].idx, q = queries[k];
            for (uint32_t h = 0; h < hcnt[k]; h++) tg[h] = hits[(size_t)k * M + h].t;
            const int32_t ms = uco_min_score_q(p, q, (int)(db->off[q + 1] - db->off[q]), dbres);
            uco_simd_align_query(db, q, tg, hcnt[k], p, ms, aln_out ? aln_out + (size_t)k * M : buf);
        }
        free(buf); free(tg);
    }
    const double t2 = now_s();
    seconds[0] = t1 - t0; seconds[1] = t2 - t1;
    free(qo);
    if (!hit_out) free(hits);
    if (!cnt_out) free(hcnt);
    return pairs;
}

/* E5/E6 of a (query, target) pair list sorted by query (the pre-step's (centre, member) pairs): one uco_simd_align_query per query group, groups in
 * parallel.  out[np] receives the records in list order.  tools/oracle_at_size.py --workflow uses it for the linear-time pre-step at full size. */
void uco_simd_align_pairs(const uco_db *db, const uint32_t *pairs, uint64_t np, const uco_params *p, int threads, uco_aln *out) {
    if (!np) return;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    uint64_t ng = 0;
    uint64_t *gb = (uint64_t *)malloc((np + 1) * sizeof(uint64_t));
    for (uint64_t k = 0; k < np; k++) if (k == 0 || pairs[2 * k] != pairs[2 * (k - 1)]) gb[ng++] = k;
    gb[ng] = np;
    const uint64_t dbres = db->off[db->n];
#pragma omp parallel
    {
        uint32_t *tg = NULL; size_t cap = 0;
#pragma omp for schedule(dynamic, 8)
        for (int64_t g = 0; g < (int64_t)ng; g++) {
            const uint64_t b = gb[g], e = gb[g + 1];
            const uint32_t q = pairs[2 * b];
            if (e - b > cap) { cap = (size_t)(e - b) * 2; tg = (uint32_t *)realloc(tg, cap * sizeof(uint32_t)); }
            for (uint64_t k = b; k < e; k++) tg[k - b] = pairs[2 * k + 1];
            const int32_t ms = uco_min_score(p, (int)(db->off[q + 1] - db->off[q]), dbres);
            uco_simd_align_query(db, q, tg, (uint32_t)(e - b), p, ms, out + b);
        }
        free(tg);
    }
    free(gb);
}
