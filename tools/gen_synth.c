/* gen_synth — deterministic synthetic proteome-set generator (SURVEY.md §8(d) "synthetic inputs").
 *
 * Writes an MMseqs/Foldseek-style sequence DB exactly as `unicore createdb` leaves it on disk
 * (reference: src/modules/createdb.rs:100-108,158-166; format witness
 * src/seq/create_gene_specific_fasta.rs:9-36):
 *     <out>        AA track      entries  "SEQ\n\0"
 *     <out>_ss     3Di track     entries  "SEQ\n\0"   (same keys / order / lengths)
 *     <out>_h      headers       entries  "unicore_<md5(aa)[:10]>\n\0"
 *     <out>{,_ss,_h}.index       "key\toffset\tlength\n"  (length includes "\n\0")
 *     <out>{,_ss,_h}.dbtype      4-byte LE int (0 = AA, 0 = 3Di stored as AA letters, 12 = generic)
 *     <out>.lookup               "key\tname\t0\n"
 *     <out>.map                  "name\tspecies\theader\n"   (createdb.rs:108)
 *
 * Family model (all draws from SplitMix64 streams keyed by (seed, object ids) so bytes are identical on
 * every machine with the same libm): F ancestral genes with equal-length (AA,3Di) tracks,
 * length ~ round(LogNormal(5.45,0.76)*len_scale) clipped to [lmin,2000]; "core" families present per
 * proteome with p=0.95, accessory with p_f~U[0.05,0.6]; 5% paralog copies; 5% proteome-private
 * singletons; members = ancestor + per-site substitutions (3Di rate U[0.05,0.35], AA rate U[0.1,0.6]),
 * indels 0.01/site (geometric, mean 3), 10% truncated to 50-79% length; 0.5% exact duplicates of the
 * previous proteome's member (exercises the md5 de-dup + multi-species map lines).
 *
 * usage: gen_synth <out_prefix> <n_proteomes> <seed> [n_families=6000] [len_scale=1.0]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const char LET[21] = "ACDEFGHIKLMNPQRSTVWY";
/* AA background fitted to example/data, order ACDEFGHIKLMNPQRSTVWY */
static const double BG_AA[20] = {.077, .008, .059, .081, .043, .065, .016, .068, .073, .093,
                                 .022, .044, .038, .033, .057, .067, .049, .067, .008, .033};
/* fixed skewed 3Di-like background (real 3Di background unknown here; documented stand-in) */
static const double BG_3DI[20] = {.040, .035, .110, .020, .030, .030, .025, .035, .025, .075,
                                  .015, .035, .085, .060, .040, .065, .020, .190, .030, .035};

typedef struct { uint64_t s; } rng_t;
static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static uint64_t rng_next(rng_t *r) { r->s += 0x9E3779B97F4A7C15ULL; return mix64(r->s); }
static rng_t rng_key(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
    rng_t r; r.s = mix64(seed ^ mix64(a * 0xD6E8FEB86659FD93ULL + mix64(b * 0xA0761D6478BD642FULL + mix64(c))));
    return r;
}
static double rng_u01(rng_t *r) { return (double)(rng_next(r) >> 11) / 9007199254740992.0; }
static double rng_normal(rng_t *r) {
    double u1 = rng_u01(r), u2 = rng_u01(r);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static int draw_letter(rng_t *r, const double *bg) {
    double u = rng_u01(r), c = 0;
    for (int i = 0; i < 20; i++) { c += bg[i]; if (u < c) return i; }
    return 19;
}

/* ---- MD5 (RFC 1321), compact ---- */
static void md5(const uint8_t *msg, size_t len, uint8_t out[16]) {
    static const uint32_t K[64] = {
        0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,
        0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
        0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,
        0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
        0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,
        0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
        0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,
        0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391};
    static const int S[64] = {7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22,5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20,
                              4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23,6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21};
    uint32_t a0 = 0x67452301, b0 = 0xefcdab89, c0 = 0x98badcfe, d0 = 0x10325476;
    size_t nl = ((len + 8) / 64 + 1) * 64;
    uint8_t *m = (uint8_t *)calloc(nl, 1);
    memcpy(m, msg, len); m[len] = 0x80;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) m[nl - 8 + i] = (uint8_t)(bits >> (8 * i));
    for (size_t off = 0; off < nl; off += 64) {
        uint32_t w[16];
        for (int i = 0; i < 16; i++)
            w[i] = (uint32_t)m[off + 4 * i] | ((uint32_t)m[off + 4 * i + 1] << 8) |
                   ((uint32_t)m[off + 4 * i + 2] << 16) | ((uint32_t)m[off + 4 * i + 3] << 24);
        uint32_t A = a0, B = b0, C = c0, D = d0;
        for (int i = 0; i < 64; i++) {
            uint32_t F; int g;
            if (i < 16) { F = (B & C) | (~B & D); g = i; }
            else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) % 16; }
            else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) % 16; }
            else { F = C ^ (B | ~D); g = (7 * i) % 16; }
            F = F + A + K[i] + w[g];
            A = D; D = C; C = B;
            B = B + ((F << S[i]) | (F >> (32 - S[i])));
        }
        a0 += A; b0 += B; c0 += C; d0 += D;
    }
    free(m);
    uint32_t h[4] = {a0, b0, c0, d0};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(h[i] >> (8 * j));
}

typedef struct { int len; uint8_t *aa, *ss; double r3, ra; } gene_t;

static int draw_len(rng_t *r, double scale, int lmin) {
    double l = exp(5.45 + 0.76 * rng_normal(r)) * scale;
    int L = (int)floor(l + 0.5);
    if (L < lmin) L = lmin;
    if (L > 2000) L = 2000;
    return L;
}
static void make_ancestor(gene_t *g, rng_t *r, double scale, int lmin) {
    g->len = draw_len(r, scale, lmin);
    g->aa = (uint8_t *)malloc(g->len); g->ss = (uint8_t *)malloc(g->len);
    for (int i = 0; i < g->len; i++) { g->aa[i] = (uint8_t)draw_letter(r, BG_AA); g->ss[i] = (uint8_t)draw_letter(r, BG_3DI); }
}
/* mutate ancestor into member; returns length; out buffers must hold 2*len+64 */
static int make_member(const gene_t *anc, rng_t *r, double r3, double ra, uint8_t *oa, uint8_t *os, int lmin) {
    int n = 0;
    for (int i = 0; i < anc->len; i++) {
        double u = rng_u01(r);
        if (u < 0.005) { /* deletion, geometric mean 3 */
            while (i + 1 < anc->len && rng_u01(r) < 2.0 / 3.0) i++;
            continue;
        }
        if (u < 0.010) { /* insertion before this site */
            do { oa[n] = (uint8_t)draw_letter(r, BG_AA); os[n] = (uint8_t)draw_letter(r, BG_3DI); n++; }
            while (rng_u01(r) < 2.0 / 3.0 && n < 2 * anc->len);
        }
        oa[n] = (rng_u01(r) < ra) ? (uint8_t)draw_letter(r, BG_AA) : anc->aa[i];
        os[n] = (rng_u01(r) < r3) ? (uint8_t)draw_letter(r, BG_3DI) : anc->ss[i];
        n++;
    }
    if (rng_u01(r) < 0.10) { /* truncation to 50-79% */
        int keep = (int)(n * (0.50 + 0.29 * rng_u01(r)));
        if (keep < lmin) keep = n < lmin ? n : lmin;
        if (rng_u01(r) < 0.5) { memmove(oa, oa + (n - keep), keep); memmove(os, os + (n - keep), keep); }
        n = keep;
    }
    if (n < 2) { oa[0] = oa[1] = 0; os[0] = os[1] = 0; n = 2; }
    return n;
}

/* open-addressing set of 40-bit name hashes → first key */
typedef struct { uint64_t *k; uint32_t *v; size_t cap; } nameset_t;
static long nameset_get_or_put(nameset_t *s, uint64_t h, uint32_t v) {
    size_t i = (size_t)(mix64(h) & (s->cap - 1));
    while (s->k[i] != UINT64_MAX) { if (s->k[i] == h) return (long)s->v[i]; i = (i + 1) & (s->cap - 1); }
    s->k[i] = h; s->v[i] = v; return -1;
}

/* ---- r06: the same bytes, generated on several threads ------------------------------------------------------------------------------
 * Every (proteome, family) has its own RNG stream, so the members of different proteomes are independent - except for two things the
 * sequential generator carried from proteome to proteome: (a) "the last first-copy member of family f" (the source of the 0.5 % exact
 * duplicates; whether one exists also decides whether the duplicate draw is consumed at all) and (b) the name set that drops a protein
 * already in the database.  (a) is a function of the per-(p, f) presence / duplicate draws alone: a cheap sequential PLAN pass replays
 * just those draws and records, per (p, f), the number of copies, whether the first copy is a duplicate and of which earlier proteome's
 * member (that member is regenerated from its own stream where it is needed).  Then the proteomes of a batch are generated in parallel
 * into memory (letters, md5 names), and (b) + all file output stay sequential, in proteome order.  Checked byte for byte against the
 * sequential generator of r01-r05 (all eleven files; 3 ... 200 proteomes, 40 / 60 / 6000 families, length scales 0.6 / 0.7 / 1, seven seeds
 * incl. every seed the tests and the bench use); every committed golden under tests/golden/ that was computed on a generated database pins
 * it as well (a changed byte changes their sha256).  UC_GEN_THREADS overrides the thread count (default: online CPUs, at most 16). */
#include <pthread.h>
#include <unistd.h>

typedef struct {
    uint8_t copies, dup, had_last;   /* first copy: exact duplicate of (src_p, f)'s first copy; had_last: a last member existed when (p, f) drew */
    int32_t src_p;
} pf_plan_t;

typedef struct {                     /* one generated proteome, in member order */
    int n_members;
    int *len;                        /* per member */
    char *aa, *ss;                   /* concatenated letters */
    size_t *off;                     /* start of member i in aa / ss */
    char (*name)[20];                /* "unicore_<10 hex>" */
    uint64_t *h40;
    size_t cap_members, cap_res;
} proteome_t;

typedef struct {
    uint64_t seed; int F, lmin; double scale;
    const gene_t *anc; const pf_plan_t *plan;   /* plan[p * F + f] */
} gen_ctx_t;

/* the first-copy member of (p, f), generated from its own stream exactly as the sequential generator did (not a duplicate itself) */
static int gen_first_copy(const gen_ctx_t *cx, int p, int f, uint8_t *oa, uint8_t *os, rng_t *r_out) {
    const pf_plan_t *pl = &cx->plan[(size_t)p * cx->F + f];
    rng_t r = rng_key(cx->seed, 2, (uint64_t)p, (uint64_t)f);
    (void)rng_u01(&r);                               /* presence (was < pf: the member exists) */
    (void)rng_u01(&r);                               /* paralog draw */
    if (pl->had_last) (void)rng_u01(&r);             /* duplicate draw (was >= 0.005) */
    double r3 = 0.05 + 0.30 * rng_u01(&r), ra = 0.10 + 0.50 * rng_u01(&r);
    int n = make_member(&cx->anc[f], &r, r3, ra, oa, os, cx->lmin);
    if (r_out) *r_out = r;
    return n;
}

static void prot_push(proteome_t *pr, const uint8_t *oa, const uint8_t *os, int n) {
    if ((size_t)pr->n_members + 1 > pr->cap_members) {
        pr->cap_members = pr->cap_members ? pr->cap_members * 2 : 4096;
        pr->len = (int *)realloc(pr->len, pr->cap_members * sizeof(int));
        pr->off = (size_t *)realloc(pr->off, pr->cap_members * sizeof(size_t));
        pr->name = (char (*)[20])realloc(pr->name, pr->cap_members * 20);
        pr->h40 = (uint64_t *)realloc(pr->h40, pr->cap_members * 8);
    }
    const size_t o = pr->n_members ? pr->off[pr->n_members - 1] + (size_t)pr->len[pr->n_members - 1] : 0;
    if (o + (size_t)n > pr->cap_res) {
        pr->cap_res = (o + (size_t)n) * 2 + 65536;
        pr->aa = (char *)realloc(pr->aa, pr->cap_res); pr->ss = (char *)realloc(pr->ss, pr->cap_res);
    }
    for (int i = 0; i < n; i++) { pr->aa[o + i] = LET[oa[i]]; pr->ss[o + i] = LET[os[i]]; }
    uint8_t dg[16]; md5((const uint8_t *)pr->aa + o, (size_t)n, dg);
    snprintf(pr->name[pr->n_members], 20, "unicore_%02x%02x%02x%02x%02x", dg[0], dg[1], dg[2], dg[3], dg[4]);
    pr->h40[pr->n_members] = ((uint64_t)dg[0] << 32) | ((uint64_t)dg[1] << 24) | ((uint64_t)dg[2] << 16) | ((uint64_t)dg[3] << 8) | dg[4];
    pr->len[pr->n_members] = n; pr->off[pr->n_members] = o;
    pr->n_members++;
}

static void gen_proteome(const gen_ctx_t *cx, int p, proteome_t *pr) {
    uint8_t *oa = (uint8_t *)malloc(8192), *os = (uint8_t *)malloc(8192);
    pr->n_members = 0;
    for (int f = 0; f < cx->F; f++) {
        const pf_plan_t *pl = &cx->plan[(size_t)p * cx->F + f];
        if (!pl->copies) continue;
        rng_t r;
        int n;
        if (pl->dup) {
            n = gen_first_copy(cx, pl->src_p, f, oa, os, NULL);
            r = rng_key(cx->seed, 2, (uint64_t)p, (uint64_t)f);
            (void)rng_u01(&r); (void)rng_u01(&r); (void)rng_u01(&r);      /* presence, paralog, duplicate (< 0.005) */
        } else {
            n = gen_first_copy(cx, p, f, oa, os, &r);
        }
        prot_push(pr, oa, os, n);
        if (pl->copies == 2) {                                             /* paralog: more diverged, the same stream goes on */
            double r3 = 0.05 + 0.30 * rng_u01(&r), ra = 0.10 + 0.50 * rng_u01(&r);
            r3 = r3 * 0.5 + 0.25; ra = ra * 0.5 + 0.40;
            n = make_member(&cx->anc[f], &r, r3, ra, oa, os, cx->lmin);
            prot_push(pr, oa, os, n);
        }
    }
    const int nsingle = (int)(pr->n_members * 0.05 + 0.5);               /* proteome-private singletons */
    for (int c = 0; c < nsingle; c++) {
        gene_t g; rng_t rs = rng_key(cx->seed, 3, (uint64_t)p, (uint64_t)c);
        make_ancestor(&g, &rs, cx->scale, cx->lmin);
        prot_push(pr, g.aa, g.ss, g.len);
        free(g.aa); free(g.ss);
    }
    free(oa); free(os);
}

typedef struct { const gen_ctx_t *cx; proteome_t *prots; int p0, np; volatile int *next; pthread_mutex_t *mu; } worker_t;
static void *worker_main(void *arg) {
    worker_t *w = (worker_t *)arg;
    for (;;) {
        pthread_mutex_lock(w->mu);
        const int k = (*w->next)++;
        pthread_mutex_unlock(w->mu);
        if (k >= w->np) break;
        gen_proteome(w->cx, w->p0 + k, &w->prots[k]);
    }
    return NULL;
}

/* buffered output with cheap integer formatting (five index / lookup lines per sequence: fprintf was a third of the run time) */
typedef struct { FILE *fp; char *buf; size_t n, cap; } out_t;
static void out_flush(out_t *o) { if (o->n) fwrite(o->buf, 1, o->n, o->fp); o->n = 0; }
static void out_bytes(out_t *o, const char *s, size_t n) {
    if (o->n + n > o->cap) { out_flush(o); if (n > o->cap) { fwrite(s, 1, n, o->fp); return; } }
    memcpy(o->buf + o->n, s, n); o->n += n;
}
static void out_ch(out_t *o, char c) { out_bytes(o, &c, 1); }
static void out_u64(out_t *o, unsigned long long v) {
    char t[24]; int k = 24;
    do { t[--k] = (char)('0' + v % 10); v /= 10; } while (v);
    out_bytes(o, t + k, (size_t)(24 - k));
}
static void out_u5(out_t *o, unsigned v) {       /* %05d */
    char t[16]; int k = 16;
    do { t[--k] = (char)('0' + v % 10); v /= 10; } while (v);
    while (16 - k < 5) t[--k] = '0';
    out_bytes(o, t + k, (size_t)(16 - k));
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <out_prefix> <n_proteomes> <seed> [n_families=6000] [len_scale=1.0]\n", argv[0]); return 2; }
    const char *out = argv[1];
    int P = atoi(argv[2]);
    uint64_t seed = strtoull(argv[3], NULL, 0);
    int F = argc > 4 ? atoi(argv[4]) : 6000;
    double scale = argc > 5 ? atof(argv[5]) : 1.0;
    int lmin = scale >= 1.0 ? 50 : 30;
    int ncore = F / 4;
    if (P < 1 || F < 4) { fprintf(stderr, "bad args\n"); return 2; }

    gene_t *anc = (gene_t *)calloc(F, sizeof(gene_t));
    double *pf = (double *)malloc(sizeof(double) * F);
    for (int f = 0; f < F; f++) {
        rng_t r = rng_key(seed, 1, (uint64_t)f, 0);
        make_ancestor(&anc[f], &r, scale, lmin);
        pf[f] = f < ncore ? 0.95 : 0.05 + 0.55 * rng_u01(&r);
    }
    /* plan pass: the presence / paralog / duplicate draws of every (p, f), in the sequential generator's order of events per family */
    pf_plan_t *plan = (pf_plan_t *)calloc((size_t)P * F, sizeof(pf_plan_t));
    for (int f = 0; f < F; f++) {
        int has_last = 0, last_src = -1;
        for (int p = 0; p < P; p++) {
            pf_plan_t *pl = &plan[(size_t)p * F + f];
            rng_t r = rng_key(seed, 2, (uint64_t)p, (uint64_t)f);
            if (!(rng_u01(&r) < pf[f])) continue;
            pl->copies = 1;
            if (rng_u01(&r) < 0.05) pl->copies = 2;
            pl->had_last = (uint8_t)has_last;
            if (has_last && rng_u01(&r) < 0.005) { pl->dup = 1; pl->src_p = last_src; }
            else { has_last = 1; last_src = p; }
        }
    }

    char path[4096];
    FILE *fa, *fs, *fh, *ia, *is, *ih, *lk, *mp;
#define OPEN(fp, suffix) do { snprintf(path, sizeof path, "%s%s", out, suffix); fp = fopen(path, "wb"); if (!fp) { perror(path); return 3; } } while (0)
    OPEN(fa, ""); OPEN(fs, "_ss"); OPEN(fh, "_h"); OPEN(ia, ".index"); OPEN(is, "_ss.index"); OPEN(ih, "_h.index");
    OPEN(lk, ".lookup"); OPEN(mp, ".map");
    out_t o_fa = {fa, 0, 0, 0}, o_fs = {fs, 0, 0, 0}, o_fh = {fh, 0, 0, 0}, o_ia = {ia, 0, 0, 0}, o_is = {is, 0, 0, 0}, o_ih = {ih, 0, 0, 0}, o_lk = {lk, 0, 0, 0}, o_mp = {mp, 0, 0, 0};
    out_t *outs[8] = {&o_fa, &o_fs, &o_fh, &o_ia, &o_is, &o_ih, &o_lk, &o_mp};
    for (int i = 0; i < 8; i++) { outs[i]->cap = (size_t)4 << 20; outs[i]->buf = (char *)malloc(outs[i]->cap); }

    nameset_t ns; ns.cap = 1; while (ns.cap < (size_t)P * (size_t)F * 2 + 1024) ns.cap <<= 1;
    ns.k = (uint64_t *)malloc(ns.cap * 8); ns.v = (uint32_t *)malloc(ns.cap * 4);
    memset(ns.k, 0xFF, ns.cap * 8);

    int T = 1;
    {
        long nc = sysconf(_SC_NPROCESSORS_ONLN);
        T = nc > 16 ? 16 : (nc < 1 ? 1 : (int)nc);
        const char *e = getenv("UC_GEN_THREADS");
        if (e && atoi(e) > 0) T = atoi(e);
        if (T > P) T = P;
    }
    const int BATCH = 2 * T > 16 ? 2 * T : 16;             /* proteomes generated (in parallel) before their output is written (in order) */
    proteome_t *prots = (proteome_t *)calloc((size_t)BATCH, sizeof(proteome_t));
    gen_ctx_t cx = {seed, F, lmin, scale, anc, plan};
    uint32_t nkeys = 0; uint64_t offa = 0, offh = 0, nres = 0, nmap = 0;

    for (int p0 = 0; p0 < P; p0 += BATCH) {
        const int np = P - p0 < BATCH ? P - p0 : BATCH;
        volatile int next = 0;
        pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
        worker_t w = {&cx, prots, p0, np, &next, &mu};
        pthread_t th[64];
        const int nt = T < np ? T : np;
        for (int t = 1; t < nt; t++) pthread_create(&th[t], NULL, worker_main, &w);
        worker_main(&w);
        for (int t = 1; t < nt; t++) pthread_join(th[t], NULL);
        for (int k = 0; k < np; k++) {
            const proteome_t *pr = &prots[k];
            char species[64]; const int sl = snprintf(species, sizeof species, "synth_%04d", p0 + k);
            for (int m = 0; m < pr->n_members; m++) {
                const char *name = pr->name[m];
                const int hl = (int)strlen(name), n = pr->len[m];
                /* "%s\t%s\t%s_g%05d\n" */
                out_bytes(&o_mp, name, (size_t)hl); out_ch(&o_mp, '\t'); out_bytes(&o_mp, species, (size_t)sl); out_ch(&o_mp, '\t');
                out_bytes(&o_mp, species, (size_t)sl); out_bytes(&o_mp, "_g", 2); out_u5(&o_mp, (unsigned)m); out_ch(&o_mp, '\n'); nmap++;
                if (nameset_get_or_put(&ns, pr->h40[m], nkeys) >= 0) continue; /* identical protein already in DB (createdb.rs:107) */
                out_bytes(&o_fa, pr->aa + pr->off[m], (size_t)n); out_bytes(&o_fa, "\n\0", 2);
                out_bytes(&o_fs, pr->ss + pr->off[m], (size_t)n); out_bytes(&o_fs, "\n\0", 2);
                out_u64(&o_ia, nkeys); out_ch(&o_ia, '\t'); out_u64(&o_ia, offa); out_ch(&o_ia, '\t'); out_u64(&o_ia, (unsigned long long)n + 2); out_ch(&o_ia, '\n');
                out_u64(&o_is, nkeys); out_ch(&o_is, '\t'); out_u64(&o_is, offa); out_ch(&o_is, '\t'); out_u64(&o_is, (unsigned long long)n + 2); out_ch(&o_is, '\n');
                offa += (uint64_t)n + 2;
                out_bytes(&o_fh, name, (size_t)hl); out_bytes(&o_fh, "\n\0", 2);
                out_u64(&o_ih, nkeys); out_ch(&o_ih, '\t'); out_u64(&o_ih, offh); out_ch(&o_ih, '\t'); out_u64(&o_ih, (unsigned long long)hl + 2); out_ch(&o_ih, '\n');
                offh += (uint64_t)hl + 2;
                out_u64(&o_lk, nkeys); out_ch(&o_lk, '\t'); out_bytes(&o_lk, name, (size_t)hl); out_bytes(&o_lk, "\t0\n", 3);
                nkeys++; nres += (uint64_t)n;
            }
        }
    }
    for (int i = 0; i < 8; i++) { out_flush(outs[i]); fclose(outs[i]->fp); }
    const int32_t dt_aa = 0, dt_h = 12;
    const char *sfx[3] = {".dbtype", "_ss.dbtype", "_h.dbtype"};
    for (int i = 0; i < 3; i++) {
        FILE *fp; OPEN(fp, sfx[i]);
        fwrite(i == 2 ? &dt_h : &dt_aa, 4, 1, fp); fclose(fp);
    }
    fprintf(stderr, "gen_synth: %u sequences, %llu residues, %llu map lines -> %s (%d threads)\n", nkeys,
            (unsigned long long)nres, (unsigned long long)nmap, out, T);
    return 0;
}
