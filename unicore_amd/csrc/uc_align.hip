// uc_align.hip — stage E5/E6 orchestration, device-resident.
// The prefilter leaves its hit lists in HBM; everything between them and the accepted edge list stays on
// the GPU: the planner sorts the pairs by (length class, query, target length), cuts them into workgroup
// tasks and gathers the kernel inputs; the gates (E-value on the corrected score, coverage) are small
// kernels with scan-based compaction.  The host only sees a handful of counters and, at the end, the
// accepted edges (stands for Foldseek's structurealign result handling; spec UC-1 E5/E6).
// Work that is never launched: the reversed-query pass of pairs below the E-value threshold (UC-1.1),
// and one of the two DPs of a mutual hit (q,t)/(t,q) in the forward, reversed-query, start and traceback passes
// whenever the result of the other orientation is provably the transposed one.
#include <mutex>
#include <hip/hip_runtime.h>

#include <cmath>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "uc_engine.h"

namespace uc {

constexpr int SW_PK_OVF_HOST = 0x7C00 - 256;   // == SW_PK_OVF of uc_sw_pk_impl.hpp

// scratch for the traceback-byte matrices of one batch (MODE 7): up to 64 GiB (r06; 144 GiB until r05), never more than 55 % of what
// is free right now (plus what the scratch buffer already holds); UC_TB_BUDGET_MB overrides (tests).  Why 64: taking device memory costs 25-40 ms per GiB
// whenever the box's memory is not warm for this process (one 108 GiB buffer: 4.2 s of a 14 s call at 500 proteomes in the first process on a fresh box,
// profiles/r06/c4_p500_alloc.txt), while the batches a smaller buffer adds cost a launch tail each (r05: halving the batches of nominal configs[3] cost 3 %
// of the pass)
static unsigned long long tb_budget_bytes(size_t already_held) {
    if (const char *e = getenv("UC_TB_BUDGET_MB")) return std::max<unsigned long long>(1, strtoull(e, nullptr, 10)) << 20;
    size_t free_b = 0, total_b = 0;
    unsigned long long cap = 64ull << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        cap = std::min<unsigned long long>(cap, (unsigned long long)((free_b + already_held) * 0.55));
    return std::max<unsigned long long>(cap, 256ull << 20);
}

namespace {

// Length classes.  Table 0: int32 kernel for everything (16 systolic classes + the row-blocked long-query kernel).  Table 1: the packed
// 16-bit kernel for all systolic classes: queries <= 2048 rows in 28 classes with even R (~6R + 60 live registers).
constexpr int MAXCLS = 28;
struct ClassTable {
    int n;                 // systolic classes; index n = long queries (> cap[n-1] rows): uc_sw_long.hip
    int cap[MAXCLS], G[MAXCLS], R[MAXCLS], pk[MAXCLS];
    uint32_t tcap[MAXCLS]; // pairs per workgroup task
};
// pairs per task: the long-query classes have few queries, so their pair lists are cut finer to keep every
// CU busy (rebuilding the LDS profile per task is negligible against >= 16 long alignments)
const ClassTable h_tab[4] = {
    {16,
     {64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1280, 1536, 1792, 2048},
     {16, 16, 16, 16, 16, 16, 16, 16, 32, 32, 32, 32, 64, 64, 64, 64},
     {4, 8, 12, 16, 20, 24, 28, 32, 20, 24, 28, 32, 20, 24, 28, 32},
     {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
     {256, 256, 256, 256, 256, 256, 256, 256, 64, 64, 64, 64, 24, 24, 24, 24}},
    {28,
     {32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 448, 512, 576, 640, 704, 768, 896, 1024, 1152, 1280, 1408, 1536, 1664, 1792, 1920, 2048},
     {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 32, 32, 32, 32, 32, 32, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64},
     {2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 14, 16, 18, 20, 22, 24, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32},
     {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
     {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 48, 48, 48, 48, 48, 48, 24, 24, 24, 24}},
    // Table 2: the int32 kernel in MODE 3 (traceback statistics) carries two more register arrays: R <= 16 keeps it at
    // <= 193 VGPRs without AGPR/scratch spills up to 1024 rows (R = 24..32 needed 275-353 registers and spilled)
    {12,
     {64, 128, 192, 256, 384, 512, 768, 1024, 1280, 1536, 1792, 2048},
     {16, 16, 16, 16, 32, 32, 64, 64, 64, 64, 64, 64},
     {4, 8, 12, 16, 12, 16, 12, 16, 20, 24, 28, 32},
     {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
     {256, 256, 256, 256, 64, 64, 24, 24, 24, 24, 24, 24}},
    // Table 3 (r06): the packed kernel for SPARSE plans - the known-score passes that re-run a handful of pairs per query (MODE 4: ambiguous end rows;
    // the second MODE 6 round: mirrors that could not take their partner's start).  With one or two pairs per query a task is ONE slot: in the classes of
    // table 1 it keeps one lane group of its workgroup busy (G = 16: an eighth of the lanes) while every wave instruction is issued for all of them - those
    // passes ran at 1.5-2.1 T cells/s against 5-6 T for the dense ones (profiles/r05/sw_pass_timing_c2_c3.txt).  Here every class spans the whole wave
    // (G = 64, R = rows / 64): the one slot of a task uses all 64 lanes; the price is the longer pipeline fill (Lt + 63 steps) and fewer rows per step to
    // amortise the per-step overhead, still ~2x fewer issued instructions per sparse pair.  R < 14: two-wave workgroups (sw_pk_kernel<64, R, 4|6, 2>).
    {16,
     {128, 256, 384, 512, 640, 768, 896, 1024, 1152, 1280, 1408, 1536, 1664, 1792, 1920, 2048},
     {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64},
     {2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32},
     {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
     {8, 8, 8, 8, 8, 8, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16}},
};
__device__ __constant__ ClassTable c_tab[4];
// which table a known-score pass of the packed path plans with: sparse (table 3) when the plan holds fewer than 2.5 pairs per query of the call
// (UC_SW_SPARSE=0: always table 1, for A/B runs; results do not depend on the table - every class computes the exact DP)
static int pk_known_tab(uint32_t n_pairs, uint32_t n_queries) {
    static const bool on = !(getenv("UC_SW_SPARSE") && atoi(getenv("UC_SW_SPARSE")) == 0);
    return on && (double)n_pairs < 2.5 * (double)n_queries ? 3 : 1;
}

__device__ __forceinline__ int class_of(int lq, int tab) {
    int c = 0;
    while (c < c_tab[tab].n && c_tab[tab].cap[c] < lq) c++;
    return c;
}
__device__ __forceinline__ uint32_t task_cap(int c, int tab) { return c < c_tab[tab].n ? c_tab[tab].tcap[c] : SW_LONG_TASK_PAIRS; }   // long queries: one pair per wave (uc_sw_long.hip)

inline dim3 grid_for(uint64_t n, uint32_t cap = 16384) {
    const uint64_t b = (n + 255) / 256;
    return dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(b, cap)));
}

// key = [ class : 5 | query : 24 | 65535 - effective target length : 16 ]
__global__ void __launch_bounds__(256) plan_key_kernel(uint32_t n, const uint32_t *q, const uint32_t *t, const int32_t *qe,
                                                       const int32_t *te, const int32_t *qs, const int32_t *ts,
                                                       const uint32_t *len, int tab, uint64_t *key, uint32_t *idx,
                                                       unsigned long long *alg_bytes /* [0] bytes, [1] DP cells */) {
    unsigned long long bytes = 0, cells = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t qq = q[i];
        const uint32_t lq = len[qq];
        const uint32_t tl = te ? (uint32_t)(te[i] + 1 - (ts ? ts[i] : 0)) : len[t[i]];
        const uint32_t ql = qe ? (uint32_t)(qe[i] + 1 - (qs ? qs[i] : 0)) : lq;
        key[i] = ((uint64_t)class_of((int)lq, tab) << 40) | ((uint64_t)qq << 16) | (uint64_t)(65535u - tl);
        idx[i] = i;
        bytes += 2ull * (ql + tl) + 32;
        cells += (unsigned long long)ql * tl;
    }
    for (int o = 32; o > 0; o >>= 1) { bytes += __shfl_down(bytes, o, 64); cells += __shfl_down(cells, o, 64); }
    if ((threadIdx.x & 63) == 0 && bytes) { atomicAdd(alg_bytes, bytes); atomicAdd(alg_bytes + 1, cells); }
}

__global__ void __launch_bounds__(256) plan_gather_kernel(uint32_t n, const uint64_t *key, const uint32_t *idx, const uint32_t *t,
                                                          const int32_t *qe, const int32_t *te, const int32_t *qs, const int32_t *ts,
                                                          uint32_t *sq, uint32_t *st, int32_t *sqe, int32_t *ste, int32_t *sqs,
                                                          int32_t *sts, uint32_t *head) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint64_t k = key[i];
        const uint32_t o = idx[i];
        sq[i] = (uint32_t)(k >> 16) & 0xFFFFFFu;
        st[i] = t[o];
        if (qe) { sqe[i] = qe[o]; ste[i] = te[o]; }
        if (qs) { sqs[i] = qs[o]; sts[i] = ts[o]; }
        head[i] = (i == 0 || (key[i - 1] >> 16) != (k >> 16)) ? i : 0u;
    }
}

__global__ void __launch_bounds__(256) plan_aux_kernel(uint32_t n, const uint32_t *idx, const int32_t *aux, int32_t *saux) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) saux[i] = aux[idx[i]];
}

__global__ void __launch_bounds__(256) plan_taskflag_kernel(uint32_t n, const uint64_t *key, const uint32_t *segstart, int tab, uint32_t *flag) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        flag[i] = ((i - segstart[i]) % task_cap((int)(key[i] >> 40), tab)) == 0 ? 1u : 0u;
}

__global__ void __launch_bounds__(256) plan_taskfill_kernel(uint32_t n, const uint64_t *key, const uint32_t *flag, const uint32_t *tpos,
                                                            SwTask *tasks, uint32_t *tcls) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint64_t k = key[i];
        const uint32_t w = tpos[i];
        tasks[w].q = (uint32_t)(k >> 16) & 0xFFFFFFu;
        tasks[w].begin = i;
        tcls[w] = (uint32_t)(k >> 40);
    }
}

// count per task + an LPT sort key: [ class : 5 | ~work : 32 ], work = pairs x longest target of the task
__global__ void __launch_bounds__(256) plan_taskcount_kernel(uint32_t ntasks, uint32_t n, SwTask *tasks, const uint32_t *tcls,
                                                             const uint64_t *key, uint64_t *tkey, uint32_t *tidx) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < ntasks; k += gridDim.x * 256) {
        const uint32_t b = tasks[k].begin;
        const uint32_t cnt = (k + 1 < ntasks ? tasks[k + 1].begin : n) - b;
        tasks[k].count = cnt;
        const uint32_t work = cnt * (65535u - (uint32_t)(key[b] & 0xFFFF));
        tkey[k] = ((uint64_t)tcls[k] << 32) | (uint64_t)(0xFFFFFFFFu - work);
        tidx[k] = k;
    }
}

__global__ void __launch_bounds__(256) plan_taskgather_kernel(uint32_t ntasks, const uint32_t *tidx, const SwTask *in, SwTask *out) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < ntasks; k += gridDim.x * 256) out[k] = in[tidx[k]];
}

// bounds[c] = first task of class >= c, bounds[NB + c] = first sorted pair of class >= c   (c = 0..NB-1)
constexpr int NB = MAXCLS + 2;
__global__ void plan_bounds_kernel(uint32_t ntasks, const uint32_t *tcls, uint32_t n, const uint64_t *key, uint32_t *bounds) {
    const uint32_t c = threadIdx.x;
    if (c >= NB) return;
    uint32_t lo = 0, hi = ntasks;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (tcls[m] < c) lo = m + 1; else hi = m; }
    bounds[c] = lo;
    lo = 0; hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if ((uint32_t)(key[m] >> 40) < c) lo = m + 1; else hi = m; }
    bounds[NB + c] = lo;
}

__global__ void __launch_bounds__(256) scatter3_kernel(uint32_t n, const uint32_t *idx, const int32_t *a, const int32_t *b,
                                                       const int32_t *c, int32_t *oa, int32_t *ob, int32_t *oc) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t o = idx[i];
        oa[o] = a[i];
        if (b) { ob[o] = b[i]; oc[o] = c[i]; }
    }
}

// ---- gates --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gate_kernel(uint32_t n, const uint32_t *sq, const int32_t *s0, const int32_t *s1,
                                                   const int32_t *minscore, uint32_t qbase, uint32_t *flag) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int32_t rev = s1 ? s1[i] : 0;
        flag[i] = (s0[i] > 0 && s0[i] - rev >= minscore[sq[i] - qbase]) ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(256) pair_gather_kernel(uint32_t n, const uint32_t *flag, const uint32_t *pos, const uint32_t *sq,
                                                          const uint32_t *st, uint32_t *q1, uint32_t *t1, uint32_t *link) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint32_t w = pos[i];
        q1[w] = sq[i]; t1[w] = st[i]; link[w] = i;
    }
}
// results of a sub-plan (sorted order idx over the gathered list) back to the parent list's positions
__global__ void __launch_bounds__(256) rev_putback_kernel(uint32_t n1, const uint32_t *idx1, const uint32_t *link, const int32_t *s, int32_t *out) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n1; i += gridDim.x * 256) out[link[idx1[i]]] = s[i];
}

// ---- mutual hits: (q,t) and (t,q) have the same forward and reversed-query score when both substitution
// matrices are symmetric (the recurrence treats the two gap directions alike), so one DP serves both ----
// key = [ min(q,t) : 24 | max(q,t) : 24 | direction : 1 ]; direction 0 = the orientation with the shorter query
// (fewer DP rows -> smaller systolic group -> less pipeline fill per pair; ties: q < t)
__global__ void __launch_bounds__(256) ukey_kernel(uint32_t n, const uint32_t *q, const uint32_t *t, const uint32_t *len,
                                                   uint64_t *key, uint32_t *idx) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t a = q[i], b = t[i], lo = min(a, b), hi = max(a, b);
        const uint32_t la = len[a], lb = len[b];
        const uint32_t worse = (la > lb || (la == lb && a > b)) ? 1u : 0u;
        key[i] = ((((uint64_t)lo << 24) | hi) << 1) | worse;
        idx[i] = i;
    }
}
// over the key-sorted list: entry j is a mirror iff its predecessor is the same unordered pair
__global__ void __launch_bounds__(256) umark_kernel(uint32_t n, const uint64_t *key, const uint32_t *idx, const uint32_t *q,
                                                    const uint32_t *t, uint32_t *fq, uint32_t *ft, uint32_t *mirror, uint32_t *rep) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const uint32_t m = (j > 0 && (key[j] >> 1) == (key[j - 1] >> 1)) ? 1u : 0u;
        mirror[j] = m;
        rep[j] = m ^ 1u;
        fq[j] = q[idx[j]];
        ft[j] = t[idx[j]];
    }
}
__global__ void __launch_bounds__(256) ugather_kernel(uint32_t n, const uint32_t *rep, const uint32_t *rpos, const uint32_t *fq,
                                                      const uint32_t *ft, uint32_t *qr, uint32_t *tr, uint32_t *jrep) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        if (!rep[j]) continue;
        const uint32_t w = rpos[j];
        qr[w] = fq[j]; tr[w] = ft[j]; jrep[w] = j;
    }
}
// forward results of the representatives (plan order) -> full key-sorted list.  A mirror gets the score; its end
// position is the representative's end with the roles swapped WHEN the packed kernel saw exactly one query row
// reach the optimum (then that row's first optimal column is the overall first optimal column, so both tie-break
// orders pick the same cell).  Otherwise it gets the "end unknown" mark (-2), which the exact re-run resolves for
// the pairs that pass the E-value gate.  k < n_pk: the representative ran in a packed class (unique-row guarantee).
__global__ void __launch_bounds__(256) uscatter_kernel(uint32_t nu, uint32_t n, const uint32_t *pidx, const uint32_t *jrep,
                                                       const uint32_t *mirror, const int32_t *su, const int32_t *qeu, const int32_t *teu,
                                                       uint32_t n_pk, int ovf, int32_t *s0, int32_t *qe0, int32_t *te0) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nu; k += gridDim.x * 256) {
        const uint32_t j = jrep[pidx[k]];
        s0[j] = su[k]; qe0[j] = qeu[k]; te0[j] = teu[k];
        if (j + 1 < n && mirror[j + 1]) {
            const bool swap_ok = k < n_pk && su[k] < ovf && qeu[k] >= 0;
            s0[j + 1] = su[k];
            qe0[j + 1] = su[k] > 0 ? (swap_ok ? teu[k] : -2) : -1;
            te0[j + 1] = su[k] > 0 ? (swap_ok ? qeu[k] : -2) : -1;
        }
    }
}
// reversed-query pass: a flagged mirror whose representative is flagged too takes its value instead of running
__global__ void __launch_bounds__(256) rev_dedup_kernel(uint32_t n, const uint32_t *mirror, uint32_t *flag, uint32_t *copy) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) copy[j] = (mirror[j] && flag[j] && flag[j - 1]) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) rev_unflag_kernel(uint32_t n, const uint32_t *copy, uint32_t *flag) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) if (copy[j]) flag[j] = 0;
}
__global__ void __launch_bounds__(256) rev_copy_kernel(uint32_t n, const uint32_t *copy, int32_t *s1) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) if (copy[j]) s1[j] = s1[j - 1];
}
// algorithmic DP cells (Lq x Lt) of the listed (optionally flagged) pairs
__global__ void __launch_bounds__(256) cells_kernel(uint32_t n, const uint32_t *q, const uint32_t *t, const uint32_t *flag,
                                                    const uint32_t *len, unsigned long long *out) {
    unsigned long long c = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        if (!flag || flag[i]) c += (unsigned long long)len[q[i]] * len[t[i]];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ---- start pass shared between mutual hits ----
// The start pass of (q,t) with end cell (qe,te) and the one of its mirror (t,q) with end cell (te,qe) are transposed
// DPs.  If the packed kernel saw exactly ONE row of the representative's DP reach the optimum, both tie-break orders
// select the same cell, so the mirror's result is the representative's with the roles swapped; otherwise the mirror
// is computed on its own (second round).
__global__ void __launch_bounds__(256) sm_flag_kernel(uint32_t n2, const uint32_t *link, const uint32_t *mirror, const uint32_t *gflag,
                                                      const uint32_t *gpos, const int32_t *qe2, const int32_t *te2, const int32_t *qe0,
                                                      const int32_t *te0, uint32_t *keep, uint32_t *partner) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        const uint32_t j = link[i];
        const bool sm = mirror[j] && j > 0 && gflag[j - 1] && qe0[j - 1] >= 0 && qe2[i] == te0[j - 1] && te2[i] == qe0[j - 1];
        keep[i] = sm ? 0u : 1u;
        partner[i] = sm ? gpos[j - 1] : 0xFFFFFFFFu;
    }
}
__global__ void __launch_bounds__(256) sm_gather_kernel(uint32_t n2, const uint32_t *flag, const uint32_t *pos, const uint32_t *q2,
                                                        const uint32_t *t2, const int32_t *qe2, const int32_t *te2, uint32_t *qa,
                                                        uint32_t *ta, int32_t *qea, int32_t *tea, uint32_t *map) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint32_t w = pos[i];
        qa[w] = q2[i]; ta[w] = t2[i]; qea[w] = qe2[i]; tea[w] = te2[i]; map[w] = i;
    }
}
// results of a sub-plan (plan order k) -> natural order of the gate-passer list; uniq = the packed known-score kernel
// found every optimal cell in one row (SW_TE_UNIQUE on the column output; results of the int32 kernel never carry it)
__global__ void __launch_bounds__(256) sm_scatter_kernel(uint32_t na, const uint32_t *idx, const uint32_t *map, const int32_t *s,
                                                         const int32_t *qo, const int32_t *to, int32_t *s2,
                                                         int32_t *q2o, int32_t *t2o, uint32_t *uniq) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < na; k += gridDim.x * 256) {
        const uint32_t i = map ? map[idx[k]] : idx[k];
        const int32_t te = to[k];
        const bool u = te >= 0 && (te & SW_TE_UNIQUE) != 0;
        s2[i] = s[k]; q2o[i] = qo[k]; t2o[i] = te >= 0 ? (te & ~SW_TE_UNIQUE) : te;
        if (uniq) uniq[i] = (u && qo[k] >= 0) ? 1u : 0u;
    }
}
__global__ void __launch_bounds__(256) sm_resolve_kernel(uint32_t n2, const uint32_t *partner, const uint32_t *uniq, int32_t *s2,
                                                         int32_t *q2o, int32_t *t2o) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        const uint32_t pr = partner[i];
        if (pr == 0xFFFFFFFFu) continue;
        if (uniq[pr]) { s2[i] = s2[pr]; q2o[i] = t2o[pr]; t2o[i] = q2o[pr]; }
        else q2o[i] = -2;
    }
}
// the start pass reaches exactly the forward score: known score of entry k of a gathered sub-list
__global__ void __launch_bounds__(256) sm_score_kernel(uint32_t nb, const uint32_t *map, const uint32_t *link, const int32_t *s0, int32_t *out) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nb; k += gridDim.x * 256) out[k] = s0[link[map[k]]];
}
__global__ void __launch_bounds__(256) iota_kernel(uint32_t n, uint32_t *out) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = i;
}
// rule UC-1/L (optional): flag = the lengths of the pair leave the coverage threshold reachable (float division, as the checker's)
__global__ void __launch_bounds__(256) len_gate_kernel(uint32_t n, const uint32_t *q, const uint32_t *t, const uint32_t *len, float cov, int cov_mode,
                                                       uint32_t *flag) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float ql = (float)len[q[i]], tl = (float)len[t[i]];
        const bool ok = len[q[i]] && len[t[i]] &&
                        (cov_mode == 0 ? (ql / tl >= cov && tl / ql >= cov) : cov_mode == 1 ? (ql / tl >= cov) : (tl / ql >= cov));
        flag[i] = ok ? 1u : 0u;
    }
}
__global__ void __launch_bounds__(256) len_gate_gather_kernel(uint32_t n, const uint32_t *flag, const uint32_t *pos, const uint32_t *q, const uint32_t *t,
                                                              uint32_t *qc, uint32_t *tc, uint32_t *map) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint32_t w = pos[i];
        qc[w] = q[i]; tc[w] = t[i]; map[w] = i;
    }
}
// records of the compacted pair list -> their places in the hit-list order (the gated pairs keep the all-zero record)
__global__ void __launch_bounds__(256) len_gate_putback_kernel(uint32_t nc, const uint32_t *map, const uc_aln *src, uc_aln *dst) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nc; i += gridDim.x * 256) dst[map[i]] = src[i];
}
// algorithmic cells of the traceback pass: the box of every entry whose statistics are asked for
__global__ void __launch_bounds__(256) cells_tb_kernel(uint32_t n2, const uint32_t *eflag, const uint32_t *link, const uint32_t *idx0, const uc_aln *alns,
                                                       unsigned long long *out) {
    unsigned long long c = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256)
        if (eflag[i]) {
            const uc_aln a = alns[idx0[link[i]]];
            c += (unsigned long long)(a.qend - a.qstart + 1) * (unsigned long long)(a.tend - a.tstart + 1);
        }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// algorithmic cells of the start pass: (qEnd+1) x (tEnd+1) per gate passer
__global__ void __launch_bounds__(256) cells_box_kernel(uint32_t n, const int32_t *qe, const int32_t *te, unsigned long long *out) {
    unsigned long long c = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) c += (unsigned long long)(qe[i] + 1) * (unsigned long long)(te[i] + 1);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

__global__ void __launch_bounds__(256) gate_scatter_kernel(uint32_t n, const uint32_t *flag, const uint32_t *pos, const uint32_t *sq,
                                                           const uint32_t *st, const int32_t *qe, const int32_t *te, uint32_t *q2,
                                                           uint32_t *t2, int32_t *qe2, int32_t *te2, uint32_t *link) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint32_t w = pos[i];
        q2[w] = sq[i]; t2[w] = st[i]; qe2[w] = qe[i]; te2[w] = te[i]; link[w] = i;
    }
}

__global__ void __launch_bounds__(256) aln_basic_kernel(uint32_t n, const uint32_t *idx, const int32_t *s0, const int32_t *s1,
                                                        const int32_t *qe, const int32_t *te, const uint32_t *pass, uc_aln *out) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uc_aln a;
        a.score = s0[i]; a.score_rev = s1 ? s1[i] : 0; a.corrected = a.score - a.score_rev;
        // positions are only defined (and only exact) for pairs that pass the E-value gate
        a.qstart = -1; a.qend = pass[i] ? qe[i] : -1; a.tstart = -1; a.tend = pass[i] ? te[i] : -1;
        a.aln_len = 0; a.idents = 0; a.pass_evalue = (int32_t)pass[i]; a.accepted = 0; a.gap_opens = 0;
        out[idx[i]] = a;
    }
}

// over the start-pass results (sorted order of plan 2): start positions, coverage gate, edge flag
__global__ void __launch_bounds__(256) finalize_kernel(uint32_t n2, const uint32_t *idx2, const uint32_t *link, const uint32_t *idx0,
                                                       const uint32_t *sq2, const uint32_t *st2, const int32_t *s2, const int32_t *q2o,
                                                       const int32_t *t2o, const uint32_t *len, float cov, int cov_mode, uc_aln *alns,
                                                       uint32_t *eflag, uint32_t *mismatch) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        uc_aln &a = alns[idx0[link[idx2[i]]]];
        if (s2[i] != a.score) atomicAdd(mismatch, 1u);
        a.qstart = a.qend - q2o[i];
        a.tstart = a.tend - t2o[i];
        const float qcov = (float)(a.qend - a.qstart + 1) / (float)len[sq2[i]];
        const float tcov = (float)(a.tend - a.tstart + 1) / (float)len[st2[i]];
        const bool ok = cov_mode == 0 ? (qcov >= cov && tcov >= cov) : cov_mode == 1 ? (tcov >= cov) : (qcov >= cov);
        a.accepted = ok;
        eflag[i] = ok;
    }
}

// seq-id stage (only with --min-seq-id > 0): pairs that passed the coverage gate get their traceback statistics
constexpr uint32_t TB_BAND_MISS = 0x7FFFFFFFu;     // "the walk left the stored band" (no traceback has length 0x7fff with 0xffff identities)
__global__ void __launch_bounds__(256) tb_gather_kernel(uint32_t n2, const uint32_t *eflag, const uint32_t *epos, const uint32_t *idx2,
                                                        const uint32_t *link, const uint32_t *idx0, const uint32_t *sq2, const uint32_t *st2,
                                                        const uc_aln *alns, uint32_t *q3, uint32_t *t3, int32_t *qs3, int32_t *qe3,
                                                        int32_t *ts3, int32_t *te3, uint32_t *src3, int32_t *sc3) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        if (!eflag[i]) continue;
        const uint32_t w = epos[i], o = idx0[link[idx2[i]]];
        const uc_aln a = alns[o];
        q3[w] = sq2[i]; t3[w] = st2[i]; qs3[w] = a.qstart; qe3[w] = a.qend; ts3[w] = a.tstart; te3[w] = a.tend;
        src3[w] = i;
        sc3[w] = a.score;                                   // H of the box's end cell: where the walk over the H bytes starts
    }
}
__global__ void __launch_bounds__(256) tb_apply_kernel(uint32_t n3, const uint32_t *idx3, const uint32_t *src3, const int32_t *pack,
                                                       const int32_t *gaps, const uint32_t *idx2, const uint32_t *link, const uint32_t *idx0,
                                                       float min_seq_id, uc_aln *alns, uint32_t *eflag, uint32_t *tie, uint32_t *miss) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n3; i += gridDim.x * 256) {
        const uint32_t i2 = src3[idx3[i]];
        uc_aln &a = alns[idx0[link[idx2[i2]]]];
        const uint32_t pk = (uint32_t)pack[i];
        if (miss) {                                          // banded MODE 7: the walk left the stored band - the record is not touched, the pair is redone
            miss[i2] = pk == TB_BAND_MISS;
            if (pk == TB_BAND_MISS) continue;
        }
        a.aln_len = (int32_t)((pk >> 16) & 0x7fffu);
        a.idents = (int32_t)(pk & 0xffffu);
        if (tie) tie[i2] = pk >> 31;
        if (gaps) a.gap_opens = gaps[i] & 0x7fffffff;   // bit 31 is the gap-direction tie mark of the pass
        if (min_seq_id > 0.0f) {
            const float sid = a.aln_len > 0 ? (float)a.idents / (float)a.aln_len : 0.0f;
            const bool ok = sid >= min_seq_id;
            a.accepted = ok;
            eflag[i2] = ok;
        }
    }
}

// ---- traceback bytes (packed MODE 7) + walk --------------------------------------------------------------------
// bytes of pair p's matrix: (longer box of its slot + G + 4) steps x NL lanes (those inside the pair's band, tb_band_of) x RB row bytes; the
// slot partner is the neighbour inside the workgroup task (pairs 2i, 2i+1 of the task), see sw_pk_kernel::start_slot
__global__ void __launch_bounds__(256) tb_size_kernel(uint32_t n_pk, const uint64_t *key, const uint32_t *segstart, const int32_t *ste,
                                                      const int32_t *sts, const int32_t *sqs, const int32_t *sqe, int band, int tab, unsigned long long *size) {
    for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < n_pk; p += gridDim.x * 256) {
        const int cls = (int)(key[p] >> 40);
        const uint32_t tcap = task_cap(cls, tab), seg = segstart[p];
        const uint32_t tb = seg + (p - seg) / tcap * tcap, pp = tb + ((p - tb) ^ 1u);
        int tl = ste[p] - sts[p] + 1;
        if (pp < n_pk && segstart[pp] == seg && (pp - seg) / tcap == (p - seg) / tcap && (key[pp] >> 16) == (key[p] >> 16))
            tl = max(tl, ste[pp] - sts[pp] + 1);
        const int G = c_tab[tab].G[cls], R = c_tab[tab].R[cls], RB = 4 * ((R + 3) / 4);
        const int nl = tb_band_of(sqs[p], sqe[p], ste[p] - sts[p] + 1, G, R, band).nl;     // lanes per step inside the pair's stored band
        size[p] = (unsigned long long)(tl + G + 4) * (unsigned long long)(nl * RB);      // (the kernel's step count is rounded up to its loop trip of 2 or 4 steps)
    }
}
// one thread per pair walks its matrix of H bytes (packed MODE 7) from the end cell (oracle: traceback(), diag > F > E, a gap is left
// as soon as it can be): (alignment length << 16 | identities | tie << 31) and the number of gaps.
// The end cell's H is the pair's score; a cell next to one whose H is known differs from it by less than 128 (vertical / horizontal
// neighbours: -open .. M + open, diagonal: min score .. M, M = the largest substitution score; the host checks M + open <= 127 and
// sends everything else through the int32 pass), so its byte gives its H exactly.  Every decision the traceback takes is a comparison
// of such values:  diagonal iff H(i,j) == H(i-1,j-1) + s(i,j);  H(i,j) == F(i,j) iff some k has H(i-k,j) - open - (k-1) ext == H(i,j),
// and the search up the column ends at the first k with H(i-k,j) < H(i,j) + k ext (F <= H in every cell, so no gap that long or longer
// can reach H(i,j));  inside a gap of value f the gap was opened from the neighbour iff H(neighbour) - open == f.  E likewise along the row.
__global__ void __launch_bounds__(256) tb_walk_kernel(uint32_t p_begin, uint32_t n_pk /* pairs [p_begin, n_pk) of the plan */, const DeviceDb db, const uint64_t *key, const uint32_t *st,
                                                      const int32_t *sqs, const int32_t *sqe, const int32_t *sts, const int32_t *ste,
                                                      const int32_t *score, int open, int ext, int band,
                                                      int tab, const uint8_t *tbm, const unsigned long long *tboff, int32_t *pack,
                                                      int32_t *gaps_out) {
    __shared__ int8_t s_S3[21 * 21], s_SA[21 * 21];
    for (int k = threadIdx.x; k < 21 * 21; k += 256) { s_S3[k] = db.S3[k]; s_SA[k] = db.SA[k]; }
    __syncthreads();
    for (uint32_t p = p_begin + blockIdx.x * 256 + threadIdx.x; p < n_pk; p += gridDim.x * 256) {
        const int cls = (int)(key[p] >> 40);
        const uint32_t q = (uint32_t)(key[p] >> 16) & 0xFFFFFFu, t = st[p];
        const int G = c_tab[tab].G[cls], R = c_tab[tab].R[cls], RB = 4 * ((R + 3) / 4);
        const int qs = sqs[p], ts = sts[p];
        const TbBand bd = tb_band_of(qs, sqe[p], ste[p] - ts + 1, G, R, band);
        const bool full = bd.nl >= G;
        const unsigned long long trow = (unsigned long long)(bd.nl * RB);
        bool miss = false;           // the walk asked for a cell outside the stored band: the pair is redone with the whole box stored
        const uint8_t *m = tbm + tboff[p];
        const uint16_t *ql = db.lt + db.off[q], *tl = db.lt + db.off[t];   // 3Di | AA << 8 per residue
        // full H of cell (ii, jj), a neighbour of a cell (or gap state) whose value is within 127 of it: ref
        auto H_at = [&](int ii, int jj, int ref) -> int {
            if (ii < qs || jj < ts) return 0;
            const int lane = ii / R, stp = jj - ts + lane;
            if (!full && (stp < lane * (R + 1) - bd.dhi || stp > lane * (R + 1) + R - 1 - bd.dlo)) { miss = true; return ref; }
            const uint32_t b = m[(unsigned long long)stp * trow + (unsigned long long)((lane % bd.nl) * RB + (ii - lane * R))];
            return ref + (int)(int8_t)(uint8_t)(b - (uint32_t)ref);
        };
        int i = sqe[p], j = ste[p], state = 0;
        int h = score[p];            // state 0: H(i,j);  states 1 / 2: the value of the gap state the walk is in
        uint32_t len = 0, id = 0, gaps = 0, tie = 0;
        while (i >= qs && j >= ts && !miss) {
            if (state == 0) {
                if (h <= 0) break;                                      // H == 0
                // A traceback is mostly diagonal steps, and a step's three loads (two residues, the byte of the diagonal neighbour) depend on the path
                // taken so far, not on loaded values: the next SPEC diagonal steps are fetched together and then taken one by one from registers - one
                // memory round trip per run of up to SPEC matches instead of one per step (r05: the walk, one thread per pair on the engine's stream
                // behind every MODE 7 batch, had grown to a fifth of the pass).
                constexpr int SPEC = 8;      // (r06: 16 measured - SW + traceback kernels of configs[3] options @ 500 8,090 -> 8,230 ms: longer runs are rarer than the registers and wasted fetches cost)
                uint32_t lqv[SPEC], ltv[SPEC], bv[SPEC];                // bv: the neighbour's byte; 0x100 = outside the box (H = 0), 0x200 = outside the band
#pragma unroll
                for (int k = 0; k < SPEC; k++) {
                    const int ii = i - k, jj = j - k;
                    const bool in = ii >= qs && jj >= ts;
                    lqv[k] = in ? ql[ii] : 0u;
                    ltv[k] = in ? tl[jj] : 0u;
                    uint32_t b = 0x100u;
                    if (in && ii - 1 >= qs && jj - 1 >= ts) {
                        const int lane = (ii - 1) / R, stp = jj - 1 - ts + lane;
                        if (!full && (stp < lane * (R + 1) - bd.dhi || stp > lane * (R + 1) + R - 1 - bd.dlo)) b = 0x200u;
                        else b = m[(unsigned long long)stp * trow + (unsigned long long)((lane % bd.nl) * RB + (ii - 1 - lane * R))];
                    }
                    bv[k] = b;
                }
                bool off_diag = false;
                uint32_t lq = 0, lt = 0;
#pragma unroll
                for (int k = 0; k < SPEC; k++) {
                    if (off_diag || miss || i < qs || j < ts || h <= 0) continue;
                    lq = lqv[k]; lt = ltv[k];
                    if (bv[k] == 0x200u) { miss = true; continue; }
                    const int hd = bv[k] == 0x100u ? 0 : h + (int)(int8_t)(uint8_t)(bv[k] - (uint32_t)h);
                    const int sc = (int)s_S3[(lq & 0xffu) * 21 + (lt & 0xffu)] + (int)s_SA[(lq >> 8) * 21 + (lt >> 8)];
                    if (h == hd + sc) { len++; id += (lq >> 8) == (lt >> 8); i--; j--; h = hd; }
                    else off_diag = true;
                }
                if (!off_diag) continue;                                // the run ended on a diagonal step (or the walk is over): next run
                bool fv = false, ev = false;
                {
                    int hu = h;
                    for (int k = 1; i - k >= qs; k++) {
                        hu = H_at(i - k, j, hu);
                        if (hu == h + open + (k - 1) * ext) { fv = true; break; }
                        if (hu < h + k * ext) break;
                    }
                }
                if (fv) {   // the tie mark: H == E as well (the transposed path of a mutual hit would take the other gap first)
                    int hl = h;
                    for (int k = 1; j - k >= ts; k++) {
                        hl = H_at(i, j - k, hl);
                        if (hl == h + open + (k - 1) * ext) { ev = true; break; }
                        if (hl < h + k * ext) break;
                    }
                    tie |= ev ? 1u : 0u;
                    state = 1;
                } else state = 2;
                gaps++;
            } else if (state == 1) {                                    // gap consuming query residue i, value h = F(i,j)
                len++;
                if (i - 1 < qs) state = 0;
                else {
                    const int hu = H_at(i - 1, j, h);                   // F(i,j) = max(F(i-1,j) - ext, H(i-1,j) - open): ext .. open above h
                    if (hu - open == h) { state = 0; h = hu; } else h += ext;
                }
                i--;
            } else {                                                    // gap consuming target residue j, value h = E(i,j)
                len++;
                if (j - 1 < ts) state = 0;
                else {
                    const int hl = H_at(i, j - 1, h);
                    if (hl - open == h) { state = 0; h = hl; } else h += ext;
                }
                j--;
            }
        }
        pack[p] = miss ? (int32_t)TB_BAND_MISS : (int32_t)((len << 16) | id | (tie << 31));
        if (gaps_out) gaps_out[p] = (int32_t)gaps;
    }
}
// byte offset of the first matrix of every task of the traceback plan (the batches are cut at task boundaries)
__global__ void __launch_bounds__(256) tb_taskoff_kernel(uint32_t ntasks, const SwTask *tasks, const unsigned long long *tboff, unsigned long long *out) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < ntasks; k += gridDim.x * 256) out[k] = tboff[tasks[k].begin];
}
// accepted pairs by forward score: the packed DP of MODE 7 is only valid below its score range
__global__ void __launch_bounds__(256) tb_split_kernel(uint32_t n2, const uint32_t *flag, const uint32_t *link, const uint32_t *idx0,
                                                       const uc_aln *alns, int ovf, uint32_t *lo, uint32_t *hi) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        const bool big = flag[i] && alns[idx0[link[i]]].score >= ovf;
        lo[i] = flag[i] && !big;
        hi[i] = big;
    }
}
__global__ void __launch_bounds__(256) tb_chunk_kernel(uint32_t n2, const uint32_t *flag, const uint32_t *pos, uint32_t lo, uint32_t hi, uint32_t *out) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) out[i] = (flag[i] && pos[i] >= lo && pos[i] < hi) ? 1u : 0u;
}

// traceback statistics shared between mutual hits: the box of (t,q) is the transposed box of (q,t); if the
// representative's traceback never had to choose between the two gap directions, the mirror's path is the transposed
// path and (alignment length, identities, gaps) are the same
__global__ void __launch_bounds__(256) tbm_flag_kernel(uint32_t n2, const uint32_t *eflag, const uint32_t *link, const uint32_t *idx0,
                                                       const uint32_t *mirror, const uint32_t *gflag, const uint32_t *gpos,
                                                       const uc_aln *alns, uint32_t *run, uint32_t *partner) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        uint32_t r = eflag[i], pr = 0xFFFFFFFFu;
        if (r) {
            const uint32_t j = link[i];
            if (mirror[j] && j > 0 && gflag[j - 1]) {
                const uint32_t ir = gpos[j - 1];
                if (eflag[ir]) {
                    const uc_aln am = alns[idx0[j]], ar = alns[idx0[j - 1]];
                    if (am.qstart == ar.tstart && am.qend == ar.tend && am.tstart == ar.qstart && am.tend == ar.qend) { r = 0; pr = ir; }
                }
            }
        }
        run[i] = r;
        partner[i] = pr;
    }
}
__global__ void __launch_bounds__(256) tbm_resolve_kernel(uint32_t n2, const uint32_t *partner, const uint32_t *tie, const uint32_t *link,
                                                          const uint32_t *idx0, float min_seq_id, uc_aln *alns, uint32_t *eflag_out,
                                                          uint32_t *run2) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        const uint32_t pr = partner[i];
        uint32_t need = 0;
        if (pr != 0xFFFFFFFFu) {
            if (tie[pr]) need = 1;
            else {
                uc_aln &am = alns[idx0[link[i]]];
                const uc_aln ar = alns[idx0[link[pr]]];
                am.aln_len = ar.aln_len; am.idents = ar.idents; am.gap_opens = ar.gap_opens;
                if (min_seq_id > 0.0f) {
                    const float sid = am.aln_len > 0 ? (float)am.idents / (float)am.aln_len : 0.0f;
                    const bool ok = sid >= min_seq_id;
                    am.accepted = ok;
                    eflag_out[i] = ok;
                }
            }
        }
        run2[i] = need;
    }
}

__global__ void __launch_bounds__(256) edge_scatter_kernel(uint32_t n2, const uint32_t *eflag, const uint32_t *epos, const uint32_t *sq2,
                                                           const uint32_t *st2, uint32_t *edges) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        if (!eflag[i]) continue;
        edges[2 * epos[i]] = sq2[i];
        edges[2 * epos[i] + 1] = st2[i];
    }
}

struct MaxU32 {
    __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

}  // namespace

// ---- planner -----------------------------------------------------------------------------------------
struct SwPlan {
    uint32_t n = 0, ntasks = 0;
    DevBuf<uint64_t> key, key2;
    DevBuf<uint32_t> idx_in, idx, sq, st, head, segstart, flag, tpos, tcls, bounds;
    DevBuf<int32_t> sqe, ste, sqs, sts, saux;
    DevBuf<SwTask> tasks, tasks_in;
    DevBuf<uint64_t> tkey, tkey2;
    DevBuf<uint32_t> tidx, tidx2;
    DevBuf<unsigned long long> bytes;
    uint32_t task_base[NB] = {0}, pair_base[NB] = {0};
    uint64_t alg_bytes = 0, cells = 0;
    bool has_ends = false, has_starts = false, has_aux = false;
    int tab = 0;
    uint8_t *tbm = nullptr;                      // packed mode 7: traceback-byte matrices and per-pair offsets
    const unsigned long long *tboff = nullptr;
    int tb_band = 0;                             // ... and the half-width of the stored diagonal band (0 = the whole box)
    // the plan's device arrays back to the allocator (a plan that has run is dead weight: ~70 B per pair; the next build_plan reserves again)
    void release() {
        key.release(); key2.release(); idx_in.release(); idx.release(); sq.release(); st.release(); head.release(); segstart.release(); flag.release(); tpos.release();
        tcls.release(); bounds.release(); sqe.release(); ste.release(); sqs.release(); sts.release(); saux.release(); tasks.release(); tasks_in.release();
        tkey.release(); tkey2.release(); tidx.release(); tidx2.release(); bytes.release();
        n = ntasks = 0;
    }
};

static void scan_u32(Engine &E, DevBuf<char> &tmp, const uint32_t *in, uint32_t *out, uint32_t n, bool inclusive_max) {
    size_t tb = 0;
    if (inclusive_max) {
        UC_HIP(rocprim::inclusive_scan(nullptr, tb, in, out, (size_t)n, MaxU32(), E.stream));
        tmp.reserve(tb + 256);
        UC_HIP(rocprim::inclusive_scan(tmp.p, tb, in, out, (size_t)n, MaxU32(), E.stream));
    } else {
        UC_HIP(rocprim::exclusive_scan(nullptr, tb, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), E.stream));
        tmp.reserve(tb + 256);
        UC_HIP(rocprim::exclusive_scan(tmp.p, tb, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), E.stream));
    }
}

// total of a 0/1 flag array given its exclusive scan
static uint32_t scan_total(Engine &E, const uint32_t *flag, const uint32_t *pos, uint32_t n) {
    if (!n) return 0;
    uint32_t a = 0, b = 0;
    UC_HIP(hipMemcpyAsync(&a, pos + (n - 1), 4, hipMemcpyDeviceToHost, E.stream));
    UC_HIP(hipMemcpyAsync(&b, flag + (n - 1), 4, hipMemcpyDeviceToHost, E.stream));
    UC_HIP(hipStreamSynchronize(E.stream));
    return a + b;
}

static void build_plan(Engine &E, SwPlan &P, DevBuf<char> &tmp, uint32_t n, const uint32_t *q, const uint32_t *t,
                       const int32_t *qe, const int32_t *te, int tab, const int32_t *qs = nullptr, const int32_t *ts = nullptr,
                       const int32_t *aux = nullptr /* one value per pair carried into plan order (known scores) */,
                       bool lpt = true /* false: tasks stay in pair order (the traceback plan is cut into byte-budgeted pair ranges) */) {
    {   // a __constant__ symbol exists once per DEVICE: upload the class tables to every device an engine plans on
        static std::mutex mu;
        static bool uploaded[64] = {};
        std::lock_guard<std::mutex> g(mu);
        if (E.device < 0 || E.device >= 64 || !uploaded[E.device]) {
            ClassTable t[4];
            memcpy(t, h_tab, sizeof t);
            UC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), t, sizeof t));
            if (E.device >= 0 && E.device < 64) uploaded[E.device] = true;
        }
    }
    P.n = n; P.ntasks = 0; P.alg_bytes = 0; P.cells = 0; P.has_ends = qe != nullptr; P.has_starts = qs != nullptr; P.tab = tab;
    P.has_aux = aux != nullptr;
    memset(P.task_base, 0, sizeof P.task_base);
    memset(P.pair_base, 0, sizeof P.pair_base);
    if (!n) return;
    hipStream_t s = E.stream;
    P.key.reserve(n); P.key2.reserve(n); P.idx_in.reserve(n); P.idx.reserve(n);
    P.sq.reserve(n); P.st.reserve(n); P.head.reserve(n); P.segstart.reserve(n); P.flag.reserve(n); P.tpos.reserve(n);
    if (qe) { P.sqe.reserve(n); P.ste.reserve(n); }
    if (qs) { P.sqs.reserve(n); P.sts.reserve(n); }
    P.bytes.reserve(2); P.bounds.reserve(2 * NB);
    UC_HIP(hipMemsetAsync(P.bytes.p, 0, 16, s));
    hipLaunchKernelGGL(plan_key_kernel, grid_for(n), dim3(256), 0, s, n, q, t, qe, te, qs, ts, E.ddb.len, tab, P.key.p, P.idx_in.p, P.bytes.p);
    size_t tb = 0;
    UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, P.key.p, P.key2.p, P.idx_in.p, P.idx.p, (size_t)n, 0u, 45u, s));
    tmp.reserve(tb + 256);
    UC_HIP(rocprim::radix_sort_pairs(tmp.p, tb, P.key.p, P.key2.p, P.idx_in.p, P.idx.p, (size_t)n, 0u, 45u, s));
    hipLaunchKernelGGL(plan_gather_kernel, grid_for(n), dim3(256), 0, s, n, P.key2.p, P.idx.p, t, qe, te, qs, ts, P.sq.p, P.st.p, P.sqe.p, P.ste.p, P.sqs.p, P.sts.p, P.head.p);
    if (aux) {
        P.saux.reserve(n);
        hipLaunchKernelGGL(plan_aux_kernel, grid_for(n), dim3(256), 0, s, n, P.idx.p, aux, P.saux.p);
    }
    scan_u32(E, tmp, P.head.p, P.segstart.p, n, true);
    hipLaunchKernelGGL(plan_taskflag_kernel, grid_for(n), dim3(256), 0, s, n, P.key2.p, P.segstart.p, tab, P.flag.p);
    scan_u32(E, tmp, P.flag.p, P.tpos.p, n, false);
    P.ntasks = scan_total(E, P.flag.p, P.tpos.p, n);
    P.tasks.reserve(P.ntasks); P.tcls.reserve(P.ntasks);
    hipLaunchKernelGGL(plan_taskfill_kernel, grid_for(n), dim3(256), 0, s, n, P.key2.p, P.flag.p, P.tpos.p, P.tasks.p, P.tcls.p);
    // longest-processing-time-first inside each class: the big tasks start first, the small ones fill the tail
    P.tasks_in.reserve(P.ntasks); P.tkey.reserve(P.ntasks); P.tkey2.reserve(P.ntasks); P.tidx.reserve(P.ntasks); P.tidx2.reserve(P.ntasks);
    hipLaunchKernelGGL(plan_taskcount_kernel, grid_for(P.ntasks), dim3(256), 0, s, P.ntasks, n, P.tasks.p, P.tcls.p, P.key2.p, P.tkey.p, P.tidx.p);
    if (lpt) {
        UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, P.tkey.p, P.tkey2.p, P.tidx.p, P.tidx2.p, (size_t)P.ntasks, 0u, 37u, s));
        tmp.reserve(tb + 256);
        UC_HIP(rocprim::radix_sort_pairs(tmp.p, tb, P.tkey.p, P.tkey2.p, P.tidx.p, P.tidx2.p, (size_t)P.ntasks, 0u, 37u, s));
        UC_HIP(hipMemcpyAsync(P.tasks_in.p, P.tasks.p, (size_t)P.ntasks * sizeof(SwTask), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(plan_taskgather_kernel, grid_for(P.ntasks), dim3(256), 0, s, P.ntasks, P.tidx2.p, P.tasks_in.p, P.tasks.p);
    }
    hipLaunchKernelGGL(plan_bounds_kernel, dim3(1), dim3(64), 0, s, P.ntasks, P.tcls.p, n, P.key2.p, P.bounds.p);
    uint32_t hb[2 * NB];
    unsigned long long bytes[2] = {0, 0};
    UC_HIP(hipMemcpyAsync(hb, P.bounds.p, sizeof hb, hipMemcpyDeviceToHost, s));
    UC_HIP(hipMemcpyAsync(bytes, P.bytes.p, 16, hipMemcpyDeviceToHost, s));
    UC_HIP(hipStreamSynchronize(s));
    UC_HIP(hipGetLastError());
    memcpy(P.task_base, hb, NB * 4);
    memcpy(P.pair_base, hb + NB, NB * 4);
    P.alg_bytes = bytes[0];
    P.cells = bytes[1];
}

// one launch per populated class; outputs are in the plan's sorted order.  Returns the number of launches.
// [t0, t1): only the tasks of that range of the plan's task list (the byte-budgeted batches of the traceback plan); skip_long: not the long-query launch
static uint64_t launch_plan(Engine &E, SwPlan &P, int mode, int32_t *os, int32_t *oqe, int32_t *ote, DevBuf<int32_t> &work,
                            const uint32_t *tb = nullptr, bool only_long = false, uint32_t t0 = 0, uint32_t t1 = 0xFFFFFFFFu, bool skip_long = false) {
    const ClassTable &tab = h_tab[P.tab];
    SwArgs a;
    a.db = E.ddb; a.tasks = P.tasks.p; a.pt = P.st.p; a.pqe = P.sqe.p; a.pte = P.ste.p;
    a.pqs = P.has_starts ? P.sqs.p : nullptr; a.pts = P.has_starts ? P.sts.p : nullptr;
    a.oscore = os; a.oqe = oqe; a.ote = ote; a.open = E.p.gap_open; a.ext = E.p.gap_ext;
    if (tb) { a.tb_diag = tb[0]; a.tb_ident = tb[1]; a.tb_open = tb[2]; a.tb_ext = tb[3]; }
    a.pscore = P.has_aux ? P.saux.p : nullptr;
    const int imode = mode >= 4 ? mode - 4 : mode;   // the int32 and long-query kernels have no known-score variant (they are exact anyway)
    if ((mode == 4 || mode == 6) && !P.has_aux) fail(UC_ERR_GENERIC, "known-score pass without scores");
    a.tbm = P.tbm; a.tboff = P.tboff; a.tb_band = P.tb_band;
    uint64_t launches = 0;
    const uint32_t gb = P.pair_base[tab.n], ngen = P.n - gb;
    uint32_t long_stride = 0;                        // queries beyond the largest systolic class: row-blocked kernel (uc_sw_long.hip)
    if (ngen) work.reserve(sw_long_work_ints(imode, ngen, E.max_len, &long_stride));
    // fork: the classes run concurrently on the auxiliary streams, largest classes first on distinct streams
    UC_HIP(hipEventRecord(E.ev_fork, E.stream));
    for (int i = 0; i < E.n_streams - 1; i++) UC_HIP(hipStreamWaitEvent(E.aux[i], E.ev_fork, 0));
    int slot = 0;
    if (ngen && !skip_long) {                        // the longest-running launch goes first
        SwArgs al = a;
        al.tasks = P.tasks.p + P.task_base[tab.n];
        launch_sw_long(imode, al, P.task_base[tab.n + 1] - P.task_base[tab.n], gb, work.p, long_stride, E.stream);
        UC_HIP(hipGetLastError());
        slot++;
        launches++;
    }
    for (int c = tab.n - 1; c >= 0 && !only_long; c--) {
        const uint32_t c0 = std::max(P.task_base[c], t0), c1 = std::min(P.task_base[c + 1], t1);
        if (c0 >= c1) continue;
        const uint32_t nt = c1 - c0;
        SwArgs ac = a;
        ac.tasks = P.tasks.p + c0;
        hipStream_t st = slot % E.n_streams == 0 ? E.stream : E.aux[slot % E.n_streams - 1];
        slot++;
        if (tab.pk[c]) launch_sw_pk_class(tab.G[c], tab.R[c], mode, ac, nt, st);
        else launch_sw_class(tab.G[c], tab.R[c], imode, ac, nt, st);
        UC_HIP(hipGetLastError());
        launches++;
    }
    for (int i = 0; i < E.n_streams - 1; i++) {   // join
        UC_HIP(hipEventRecord(E.ev_join[i], E.aux[i]));
        UC_HIP(hipStreamWaitEvent(E.stream, E.ev_join[i], 0));
    }
    return launches;
}

// pairs of the packed classes whose result is not final: ambiguous end row (qe == -2) or a value near the
// u16 range -> they are re-run by the int32 kernel
__global__ void __launch_bounds__(256) pk_flag_kernel(uint32_t n_pk, const int32_t *os, const int32_t *oqe, int ovf, int ovf_only, uint32_t *flag) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_pk; i += gridDim.x * 256)
        flag[i] = (os[i] >= ovf || (!ovf_only && oqe && oqe[i] == -2)) ? 1u : 0u;
}
// deferred variant: only the pairs that passed the E-value gate need their exact end row
__global__ void __launch_bounds__(256) amb_flag_kernel(uint32_t n2, const int32_t *qe2, uint32_t *flag) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) flag[i] = qe2[i] == -2 ? 1u : 0u;
}
__global__ void __launch_bounds__(256) amb_gather_kernel(uint32_t n2, const uint32_t *flag, const uint32_t *pos, const uint32_t *q2,
                                                         const uint32_t *t2, const uint32_t *link, const int32_t *s0, uint32_t *q3,
                                                         uint32_t *t3, uint32_t *link3, int32_t *s3) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint32_t w = pos[i];
        q3[w] = q2[i]; t3[w] = t2[i]; link3[w] = i; s3[w] = s0[link[i]];
    }
}
__global__ void __launch_bounds__(256) amb_putback_kernel(uint32_t n3, const uint32_t *idx3, const uint32_t *link3, const uint32_t *link,
                                                          const int32_t *qe3, const int32_t *te3, int32_t *qe2, int32_t *te2,
                                                          int32_t *qe0, int32_t *te0) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n3; i += gridDim.x * 256) {
        const uint32_t j = link3[idx3[i]];
        qe2[j] = qe3[i]; te2[j] = te3[i];
        qe0[link[j]] = qe3[i]; te0[link[j]] = te3[i];
    }
}
__global__ void __launch_bounds__(256) pk_gather_kernel(uint32_t n_pk, const uint32_t *flag, const uint32_t *pos, const uint32_t *sq,
                                                        const uint32_t *st, const int32_t *sqe, const int32_t *ste, uint32_t *q2,
                                                        uint32_t *t2, int32_t *qe2, int32_t *te2, uint32_t *link) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_pk; i += gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint32_t w = pos[i];
        q2[w] = sq[i]; t2[w] = st[i]; link[w] = i;
        if (sqe) { qe2[w] = sqe[i]; te2[w] = ste[i]; }
    }
}
__global__ void __launch_bounds__(256) pk_putback_kernel(uint32_t n2, const uint32_t *idx2, const uint32_t *link, const int32_t *s,
                                                         const int32_t *qe, const int32_t *te, int32_t *os, int32_t *oqe, int32_t *ote) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        const uint32_t o = link[idx2[i]];
        os[o] = s[i];
        if (oqe) { oqe[o] = qe[i]; ote[o] = te[i]; }
    }
}


struct RerunBufs {
    DevBuf<uint32_t> flag, pos, q2, t2, link;
    DevBuf<int32_t> qe2, te2, s, qe, te, sin;
};

// Work buffers of the gapped stage and of the set-cover graph build.  ONE set per engine (parked per device between
// engines): nothing in this file is process-wide, so engines on different devices - or several engines on one device,
// each driven by its own host thread as uc_cluster does for num_gpus > 1 - never share or race on a buffer.
struct AlignScratch {
    DevBuf<int32_t> d_ms, s0, qe0, te0, s1, s1c, qe2, te2, s2, q2o, t2o, work;
    DevBuf<uint32_t> gflag, gpos, q1, t1, link1, q2, t2, link, eflag, epos, mism, d_e;
    DevBuf<uint64_t> ukey, ukey2;
    DevBuf<uint32_t> uidx_in, uidx, fq, ft, mirror, rep, rpos, qr, tr, jrep, rcopy;
    DevBuf<int32_t> su, qeu, teu, s2s, q2os, t2os, qe2a, te2a, sknown;
    DevBuf<uint32_t> iota2, smkeep, smpos, partner, uniq, q2a, t2a, mapa;
    DevBuf<uint32_t> lg_q, lg_t, lg_map, lg_flag, lg_pos;   // rule UC-1/L: the compacted pair list of a batch and its records
    DevBuf<uc_aln> lg_alns;
    DevBuf<unsigned long long> d_cells;
    DevBuf<char> tmp;
    SwPlan P0, P1, P2, P2b;
    // packed-range / ambiguous-end re-runs (run_plan, fix_ambiguous_ends)
    RerunBufs rr_B, amb_B;
    SwPlan rr_P2, amb_P3;
    // traceback statistics (align: --min-seq-id / search)
    DevBuf<uint32_t> tb_q3, tb_t3, tb_src3, tb_trun, tb_tpos, tb_tpart, tb_ttie, tb_tlo, tb_thi, tb_tmiss, tb_tchunk, tb_tcpos;
    DevBuf<unsigned long long> tb_tbsize, tb_tboff, tb_toff;
    DevBuf<uint8_t> tb_tbm;
    bool tb_tbm_live = false;        // inside the batch loop of a MODE 7 call: the one time the matrices may not be given back (release_tb_matrices)
    DevBuf<int32_t> tb_qs3, tb_qe3, tb_ts3, tb_te3, tb_pack3, tb_gaps3, tb_sc3;
    SwPlan tb_P3;
    // set-cover graph build + greedy cover (set_cover_graph)
    DevBuf<uint32_t> sc_e, sc_flag, sc_pos, sc_dadj, sc_bad, sc_assign, sc_cnt, sc_work, sc_work2, sc_picks, sc_newly, sc_ctr;
    DevBuf<uint64_t> sc_m1;
    DevBuf<uint64_t> sc_key, sc_key2, sc_ukey, sc_doff;
    DevBuf<char> sc_tmp;
};
static AlignScratch &scratch_of(Engine &E) {
    if (!E.aln) E.aln = take_align_scratch(E.device);
    return *E.aln;
}

// ovf_only: re-run immediately only what saturated; ambiguous end rows (qe == -2) are left for the caller
static void run_plan(Engine &E, SwPlan &P, int mode, int32_t *os, int32_t *oqe, int32_t *ote, DevBuf<int32_t> &work,
                     DevBuf<char> &tmp, bool ovf_only = false) {
    if (!P.n) return;
    E.timed_ms_begin();
    uint64_t launches = launch_plan(E, P, mode, os, oqe, ote, work);
    double ms = E.timed_ms_end();
    const ClassTable &tab = h_tab[P.tab];
    int npk = 0;
    while (npk < tab.n && tab.pk[npk]) npk++;
    const uint32_t n_pk = P.pair_base[npk];   // packed classes form a prefix of the sorted pair list
    if (n_pk) {
        RerunBufs &B = scratch_of(E).rr_B;
        SwPlan &P2 = scratch_of(E).rr_P2;
        hipStream_t s = E.stream;
        B.flag.reserve(n_pk); B.pos.reserve(n_pk);
        hipLaunchKernelGGL(pk_flag_kernel, grid_for(n_pk), dim3(256), 0, s, n_pk, os, oqe, SW_PK_OVF_HOST, ovf_only ? 1 : 0, B.flag.p);
        scan_u32(E, tmp, B.flag.p, B.pos.p, n_pk, false);
        const uint32_t n2 = scan_total(E, B.flag.p, B.pos.p, n_pk);
        E.stats.n_pk_reruns += n2;
        if (n2) {
            B.q2.reserve(n2); B.t2.reserve(n2); B.link.reserve(n2); B.s.reserve(n2);
            if (P.has_ends) { B.qe2.reserve(n2); B.te2.reserve(n2); }
            if (oqe) { B.qe.reserve(n2); B.te.reserve(n2); }
            hipLaunchKernelGGL(pk_gather_kernel, grid_for(n_pk), dim3(256), 0, s, n_pk, B.flag.p, B.pos.p, P.sq.p, P.st.p,
                               P.has_ends ? P.sqe.p : nullptr, P.has_ends ? P.ste.p : nullptr, B.q2.p, B.t2.p, B.qe2.p, B.te2.p, B.link.p);
            build_plan(E, P2, tmp, n2, B.q2.p, B.t2.p, P.has_ends ? B.qe2.p : nullptr, P.has_ends ? B.te2.p : nullptr, 0);
            E.timed_ms_begin();
            launches += launch_plan(E, P2, mode >= 4 ? mode - 4 : mode, B.s.p, oqe ? B.qe.p : nullptr, oqe ? B.te.p : nullptr, work);
            ms += E.timed_ms_end();
            E.stats.cells_run += P2.cells;
            E.stats.n_sw_runs += P2.n;
            hipLaunchKernelGGL(pk_putback_kernel, grid_for(n2), dim3(256), 0, s, n2, P2.idx.p, B.link.p, B.s.p,
                               oqe ? B.qe.p : nullptr, oqe ? B.te.p : nullptr, os, oqe, ote);
            UC_HIP(hipGetLastError());
        }
    }
    E.stats.sw_kernel_ms += ms;
    E.stats.sw_kernel_launches += launches;
    E.stats.sw_algorithmic_bytes += P.alg_bytes;
    E.stats.cells_run += P.cells;
    E.stats.n_sw_runs += P.n;
    const bool timing = getenv("UC_TIMING") != nullptr;
    if (timing) fprintf(stderr, "unicore-cluster[timing]: sw pass mode %d: %u pairs, %llu cells, %.2f ms (%llu launches)\n", mode, P.n, (unsigned long long)P.cells, ms, (unsigned long long)launches);
}

// exact (qEnd, tEnd) by the int32 forward kernel for the gate-passing pairs the packed kernel left ambiguous
// exact (qEnd, tEnd) for the gate-passing pairs whose end is still unknown (ambiguous end row of the packed kernel, or a
// mirror that could not take its representative's end): a second forward pass that KNOWS the optimum score (packed
// MODE 4; int32 kernel for --sw-kernel i32 and for queries beyond the systolic classes)
static void fix_ambiguous_ends(Engine &E, uint32_t n2, const uint32_t *q2, const uint32_t *t2, int32_t *qe2, int32_t *te2,
                               const uint32_t *link, const int32_t *s0, int32_t *qe0, int32_t *te0, DevBuf<int32_t> &work,
                               DevBuf<char> &tmp, uint32_t n_queries) {
    RerunBufs &B = scratch_of(E).amb_B;
    SwPlan &P3 = scratch_of(E).amb_P3;
    hipStream_t s = E.stream;
    B.flag.reserve(n2); B.pos.reserve(n2);
    hipLaunchKernelGGL(amb_flag_kernel, grid_for(n2), dim3(256), 0, s, n2, qe2, B.flag.p);
    scan_u32(E, tmp, B.flag.p, B.pos.p, n2, false);
    const uint32_t n3 = scan_total(E, B.flag.p, B.pos.p, n2);
    E.stats.n_pk_reruns += n3;
    if (!n3) return;
    B.q2.reserve(n3); B.t2.reserve(n3); B.link.reserve(n3); B.s.reserve(n3); B.qe.reserve(n3); B.te.reserve(n3); B.sin.reserve(n3);
    hipLaunchKernelGGL(amb_gather_kernel, grid_for(n2), dim3(256), 0, s, n2, B.flag.p, B.pos.p, q2, t2, link, s0, B.q2.p, B.t2.p, B.link.p, B.sin.p);
    const bool pk = E.p.sw_pk != 0;
    build_plan(E, P3, tmp, n3, B.q2.p, B.t2.p, nullptr, nullptr, pk ? pk_known_tab(n3, n_queries) : 0, nullptr, nullptr, pk ? B.sin.p : nullptr);
    run_plan(E, P3, pk ? 4 : 0, B.s.p, B.qe.p, B.te.p, work, tmp);
    hipLaunchKernelGGL(amb_putback_kernel, grid_for(n3), dim3(256), 0, s, n3, P3.idx.p, B.link.p, link, B.qe.p, B.te.p, qe2, te2, qe0, te0);
    UC_HIP(hipGetLastError());
}

void free_align_scratch(AlignScratch *p) { delete p; }

// memory pressure inside the gapped stage: the traceback-byte buffer (up to 144 GiB, kept between calls because fresh memory is slow) is dead weight
// outside the batch loop of a MODE 7 call - the engine's out-of-memory handler gives it back before it gives up
bool release_tb_matrices(AlignScratch *p) {
    if (!p || p->tb_tbm_live || !p->tb_tbm.cap) return false;
    p->tb_tbm.release();
    return true;
}

// parked between engines like the prefilter's work buffers (uc_prefilter.hip)
namespace {
std::mutex g_park_mutex_aln;
AlignScratch *g_parked_aln[16] = {};
}  // namespace
void park_align_scratch(AlignScratch *p, int device) {
    if (!p) return;
    const char *e = getenv("UC_KEEP_SCRATCH");
    if (!(e && e[0] == '0') && device >= 0 && device < 16) {
        std::lock_guard<std::mutex> g(g_park_mutex_aln);
        if (!g_parked_aln[device]) { g_parked_aln[device] = p; return; }
    }
    delete p;
}
AlignScratch *take_parked_align_scratch(int device) {   // nullptr if nothing is parked (uc_release_scratch)
    if (device < 0 || device >= 16) return nullptr;
    std::lock_guard<std::mutex> g(g_park_mutex_aln);
    AlignScratch *p = g_parked_aln[device];
    g_parked_aln[device] = nullptr;
    return p;
}
AlignScratch *take_align_scratch(int device) {
    if (device >= 0 && device < 16) {
        std::lock_guard<std::mutex> g(g_park_mutex_aln);
        if (AlignScratch *p = g_parked_aln[device]) { g_parked_aln[device] = nullptr; return p; }
    }
    return new AlignScratch;
}

// ---- kernel-level entry point: arbitrary pair list from the host -----------------------------------
void Engine::sw_batch(int mode, const std::vector<PairIn> &pairs, int32_t *score, int32_t *qe, int32_t *te) {
    PressureScope ps(*this, 1);
    if (!have_db) fail(UC_ERR_ARGS, "no database loaded");
    const size_t n = pairs.size();
    if (n == 0) return;
    if (n >= (1ull << 31)) fail(UC_ERR_ARGS, "sw_batch: too many pairs");
    UC_HIP(hipSetDevice(device));
    const bool track = mode != 1;
    std::vector<uint32_t> hq(n), ht(n);
    std::vector<int32_t> hqe, hte;
    if (mode == 2) { hqe.resize(n); hte.resize(n); }
    for (size_t i = 0; i < n; i++) {
        const PairIn &x = pairs[i];
        if (x.q >= hdb.n || x.t >= hdb.n) fail(UC_ERR_ARGS, "sw_batch: sequence id out of range");
        if (mode == 2 && (x.qe < 0 || x.te < 0 || (uint32_t)x.qe >= h_len[x.q] || (uint32_t)x.te >= h_len[x.t]))
            fail(UC_ERR_ARGS, "sw_batch: end position out of range");
        hq[i] = x.q; ht[i] = x.t;
        if (mode == 2) { hqe[i] = x.qe; hte[i] = x.te; }
    }
    DevBuf<uint32_t> dq, dt;
    DevBuf<int32_t> dqe, dte, os, oq, ot, rs, rq, rt, work;
    DevBuf<char> tmp;
    dq.reserve(n); dt.reserve(n); os.reserve(n); rs.reserve(n);
    UC_HIP(hipMemcpyAsync(dq.p, hq.data(), n * 4, hipMemcpyHostToDevice, stream));
    UC_HIP(hipMemcpyAsync(dt.p, ht.data(), n * 4, hipMemcpyHostToDevice, stream));
    if (mode == 2) {
        dqe.reserve(n); dte.reserve(n);
        UC_HIP(hipMemcpyAsync(dqe.p, hqe.data(), n * 4, hipMemcpyHostToDevice, stream));
        UC_HIP(hipMemcpyAsync(dte.p, hte.data(), n * 4, hipMemcpyHostToDevice, stream));
    }
    if (track) { oq.reserve(n); ot.reserve(n); rq.reserve(n); rt.reserve(n); }
    SwPlan P;
    build_plan(*this, P, tmp, (uint32_t)n, dq.p, dt.p, dqe.p, dte.p, p.sw_pk ? 1 : 0);
    run_plan(*this, P, mode, os.p, oq.p, ot.p, work, tmp);
    hipLaunchKernelGGL(scatter3_kernel, grid_for(n), dim3(256), 0, stream, (uint32_t)n, P.idx.p, os.p, oq.p, ot.p, rs.p, rq.p, rt.p);
    UC_HIP(hipMemcpyAsync(score, rs.p, n * 4, hipMemcpyDeviceToHost, stream));
    if (track && qe) UC_HIP(hipMemcpyAsync(qe, rq.p, n * 4, hipMemcpyDeviceToHost, stream));
    if (track && te) UC_HIP(hipMemcpyAsync(te, rt.p, n * 4, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipStreamSynchronize(stream));
    UC_HIP(hipGetLastError());
}

// ---- stage E5 + E6 for queries [qbegin, qend) of the device-resident hit lists ----------------------
void Engine::align(uint32_t qbegin, uint32_t qend) {
    PressureScope ps(*this, 1);
    if (!have_db) fail(UC_ERR_ARGS, "no database loaded");
    if (qbegin > qend || qend > hdb.n) fail(UC_ERR_ARGS, "align: bad query range");
    UC_HIP(hipSetDevice(device));
    Timer tm;
    hipStream_t s = stream;
    const uint64_t dbres = evalue_residues ? evalue_residues : hdb.residues();
    // records of queries outside [qbegin,qend) must read as "not aligned" (all zero), never as leftovers of an earlier hit
    // list: the array is cleared whenever the hit lists changed since the last align (0.2 ms at 14 M records)
    d_alns.reserve(std::max<uint64_t>(n_hits, 1));
    if (!alns_valid && n_hits) UC_HIP(hipMemsetAsync(d_alns.p, 0, n_hits * sizeof(uc_aln), s));
    // E-value gate as an integer threshold per query length (host; exp() evaluated once per distinct length)
    std::vector<int32_t> ms_by_len(65536, -1), h_ms(qend - qbegin);
    if (!p.min_score_table.empty() && p.min_score_table.size() != hdb.n)
        fail(UC_ERR_ARGS, "--min-score-table holds %zu thresholds, the database has %u sequences", p.min_score_table.size(), hdb.n);
    for (uint32_t q = qbegin; q < qend; q++) {
        if (!p.min_score_table.empty()) { h_ms[q - qbegin] = p.min_score_table[q]; continue; }     // rule UC-1/E (optional): thresholds of a fitted per-query model
        int32_t &m = ms_by_len[h_len[q]];
        if (m < 0) m = min_score_for(p, (int)h_len[q], dbres);
        h_ms[q - qbegin] = m;
    }
    if (!aln) aln = take_align_scratch(device);
    AlignScratch &A = *aln;
    DevBuf<int32_t> &d_ms = A.d_ms, &s0 = A.s0, &qe0 = A.qe0, &te0 = A.te0, &s1 = A.s1, &s1c = A.s1c, &qe2 = A.qe2, &te2 = A.te2, &s2 = A.s2,
                    &q2o = A.q2o, &t2o = A.t2o, &work = A.work;
    DevBuf<uint32_t> &gflag = A.gflag, &gpos = A.gpos, &q1 = A.q1, &t1 = A.t1, &link1 = A.link1, &q2 = A.q2, &t2 = A.t2, &link = A.link,
                     &eflag = A.eflag, &epos = A.epos, &mism = A.mism, &d_e = A.d_e;
    DevBuf<uint64_t> &ukey = A.ukey, &ukey2 = A.ukey2;
    DevBuf<uint32_t> &uidx_in = A.uidx_in, &uidx = A.uidx, &fq = A.fq, &ft = A.ft, &mirror = A.mirror, &rep = A.rep, &rpos = A.rpos, &qr = A.qr,
                     &tr = A.tr, &jrep = A.jrep, &rcopy = A.rcopy;
    DevBuf<int32_t> &su = A.su, &qeu = A.qeu, &teu = A.teu, &s2s = A.s2s, &q2os = A.q2os, &t2os = A.t2os, &qe2a = A.qe2a, &te2a = A.te2a,
                    &sknown = A.sknown;
    DevBuf<uint32_t> &iota2 = A.iota2, &smkeep = A.smkeep, &smpos = A.smpos, &partner = A.partner, &uniq = A.uniq, &q2a = A.q2a, &t2a = A.t2a,
                     &mapa = A.mapa;
    DevBuf<unsigned long long> &d_cells = A.d_cells;
    d_cells.reserve(2);
    DevBuf<char> &tmp = A.tmp;
    SwPlan &P0 = A.P0, &P1 = A.P1, &P2 = A.P2, &P2b = A.P2b;
    d_ms.reserve(std::max<size_t>(h_ms.size(), 1));
    if (!h_ms.empty()) UC_HIP(hipMemcpyAsync(d_ms.p, h_ms.data(), h_ms.size() * 4, hipMemcpyHostToDevice, s));
    mism.reserve(1);
    UC_HIP(hipMemsetAsync(mism.p, 0, 4, s));

    const uint64_t CHUNK = 256ull << 20;  // pairs per device batch (~150 B of plan/result state per pair; mutual hits only share a DP inside a batch)
    for (uint32_t qa = qbegin; qa < qend;) {
        uint32_t qb = qa;
        while (qb < qend && (qb == qa || hit_off[qb + 1] - hit_off[qa] <= CHUNK)) qb++;
        const uint64_t b = hit_off[qa];
        const uint32_t n_listed = (uint32_t)(hit_off[qb] - b);
        uint32_t n = n_listed;
        const uint32_t *dq = d_hq.p + b, *dt = d_ht.p + b;
        uc_aln *alns_b = d_alns.p + b;                 // where record i of the batch's pair list goes
        if (p.len_gate && p.cov > 0.0f && n) {
            // rule UC-1/L (optional): the stage works on the pairs the length gate lets through - a compacted copy of the batch's pair list with its own
            // record array; the records return to their places in the hit-list order at the end of the batch, the gated pairs keep all-zero records
            DevBuf<uint32_t> &lgq = A.lg_q, &lgt = A.lg_t, &lgmap = A.lg_map, &lgflag = A.lg_flag, &lgpos = A.lg_pos;
            lgflag.reserve(n); lgpos.reserve(n);
            hipLaunchKernelGGL(len_gate_kernel, grid_for(n), dim3(256), 0, s, n, dq, dt, ddb.len, p.cov, p.cov_mode, lgflag.p);
            scan_u32(*this, tmp, lgflag.p, lgpos.p, n, false);
            const uint32_t nc = scan_total(*this, lgflag.p, lgpos.p, n);
            lgq.reserve(std::max<uint32_t>(nc, 1)); lgt.reserve(std::max<uint32_t>(nc, 1)); lgmap.reserve(std::max<uint32_t>(nc, 1));
            A.lg_alns.reserve(std::max<uint32_t>(nc, 1));
            hipLaunchKernelGGL(len_gate_gather_kernel, grid_for(n), dim3(256), 0, s, n, lgflag.p, lgpos.p, dq, dt, lgq.p, lgt.p, lgmap.p);
            if (nc) UC_HIP(hipMemsetAsync(A.lg_alns.p, 0, (size_t)nc * sizeof(uc_aln), s));
            UC_HIP(hipMemsetAsync(d_alns.p + b, 0, (size_t)n * sizeof(uc_aln), s));
            dq = lgq.p; dt = lgt.p; alns_b = A.lg_alns.p; n = nc;
        }
        if (n) {
            s0.reserve(n); qe0.reserve(n); te0.reserve(n); gflag.reserve(n); gpos.reserve(n);
            const int tab = p.sw_pk ? 1 : 0;
            // the directed pair list the gates work on: sorted (query, target) + position in the hit list
            const uint32_t *Lsq = nullptr, *Lst = nullptr, *Lidx = nullptr;
            const bool dedup = p.sym_dedup && p.mat_symmetric && n >= 2 && hdb.n <= (1u << 24);
            UC_HIP(hipMemsetAsync(d_cells.p, 0, 16, s));
            if (!dedup) {
                build_plan(*this, P0, tmp, n, dq, dt, nullptr, nullptr, tab);
                run_plan(*this, P0, 0, s0.p, qe0.p, te0.p, work, tmp, /*ovf_only=*/true);
                stats.cells_fwd += P0.cells;
                Lsq = P0.sq.p; Lst = P0.st.p; Lidx = P0.idx.p;
            } else {
                // mutual hits share one forward DP, computed in the orientation with the shorter query
                ukey.reserve(n); ukey2.reserve(n); uidx_in.reserve(n); uidx.reserve(n);
                fq.reserve(n); ft.reserve(n); mirror.reserve(n); rep.reserve(n); rpos.reserve(n);
                hipLaunchKernelGGL(ukey_kernel, grid_for(n), dim3(256), 0, s, n, dq, dt, ddb.len, ukey.p, uidx_in.p);
                size_t tb = 0;
                UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, ukey.p, ukey2.p, uidx_in.p, uidx.p, (size_t)n, 0u, 49u, s));
                tmp.reserve(tb + 256);
                UC_HIP(rocprim::radix_sort_pairs(tmp.p, tb, ukey.p, ukey2.p, uidx_in.p, uidx.p, (size_t)n, 0u, 49u, s));
                hipLaunchKernelGGL(umark_kernel, grid_for(n), dim3(256), 0, s, n, ukey2.p, uidx.p, dq, dt, fq.p, ft.p, mirror.p, rep.p);
                scan_u32(*this, tmp, rep.p, rpos.p, n, false);
                const uint32_t nu = scan_total(*this, rep.p, rpos.p, n);
                qr.reserve(nu); tr.reserve(nu); jrep.reserve(nu); su.reserve(nu); qeu.reserve(nu); teu.reserve(nu);
                hipLaunchKernelGGL(ugather_kernel, grid_for(n), dim3(256), 0, s, n, rep.p, rpos.p, fq.p, ft.p, qr.p, tr.p, jrep.p);
                build_plan(*this, P0, tmp, nu, qr.p, tr.p, nullptr, nullptr, tab);
                run_plan(*this, P0, 0, su.p, qeu.p, teu.p, work, tmp, /*ovf_only=*/true);
                uint32_t n_pk0 = 0;   // packed classes form a prefix of the plan's sorted pair list
                {
                    const ClassTable &ct = h_tab[P0.tab];
                    int npk = 0;
                    while (npk < ct.n && ct.pk[npk]) npk++;
                    n_pk0 = P0.pair_base[npk];
                }
                hipLaunchKernelGGL(uscatter_kernel, grid_for(nu), dim3(256), 0, s, nu, n, P0.idx.p, jrep.p, mirror.p, su.p, qeu.p, teu.p,
                                   n_pk0, SW_PK_OVF_HOST, s0.p, qe0.p, te0.p);
                hipLaunchKernelGGL(cells_kernel, grid_for(n), dim3(256), 0, s, n, fq.p, ft.p, (const uint32_t *)nullptr, ddb.len, d_cells.p);
                Lsq = fq.p; Lst = ft.p; Lidx = uidx.p;
            }
            if (p.rev_correction) {
                // spec UC-1.1: the reversed-query pass only runs for pairs whose forward score reaches the E-value
                // threshold (corrected <= score, so the others cannot pass); their score_rev is reported as 0
                s1.reserve(n);
                UC_HIP(hipMemsetAsync(s1.p, 0, (size_t)n * 4, s));
                hipLaunchKernelGGL(gate_kernel, grid_for(n), dim3(256), 0, s, n, Lsq, s0.p, (const int32_t *)nullptr, d_ms.p, qbegin, gflag.p);
                if (dedup) {
                    rcopy.reserve(n);
                    hipLaunchKernelGGL(cells_kernel, grid_for(n), dim3(256), 0, s, n, Lsq, Lst, gflag.p, ddb.len, d_cells.p + 1);
                    hipLaunchKernelGGL(rev_dedup_kernel, grid_for(n), dim3(256), 0, s, n, mirror.p, gflag.p, rcopy.p);
                    hipLaunchKernelGGL(rev_unflag_kernel, grid_for(n), dim3(256), 0, s, n, rcopy.p, gflag.p);
                }
                scan_u32(*this, tmp, gflag.p, gpos.p, n, false);
                const uint32_t n1 = scan_total(*this, gflag.p, gpos.p, n);
                if (n1) {
                    q1.reserve(n1); t1.reserve(n1); link1.reserve(n1); s1c.reserve(n1);
                    hipLaunchKernelGGL(pair_gather_kernel, grid_for(n), dim3(256), 0, s, n, gflag.p, gpos.p, Lsq, Lst, q1.p, t1.p, link1.p);
                    build_plan(*this, P1, tmp, n1, q1.p, t1.p, nullptr, nullptr, tab);
                    run_plan(*this, P1, 1, s1c.p, nullptr, nullptr, work, tmp);
                    hipLaunchKernelGGL(rev_putback_kernel, grid_for(n1), dim3(256), 0, s, n1, P1.idx.p, link1.p, s1c.p, s1.p);
                    if (!dedup) stats.cells_rev += P1.cells;
                }
                if (dedup) hipLaunchKernelGGL(rev_copy_kernel, grid_for(n), dim3(256), 0, s, n, rcopy.p, s1.p);
            }
            if (dedup) {
                unsigned long long hc[2] = {0, 0};
                UC_HIP(hipMemcpyAsync(hc, d_cells.p, 16, hipMemcpyDeviceToHost, s));
                UC_HIP(hipStreamSynchronize(s));
                stats.cells_fwd += hc[0];
                stats.cells_rev += hc[1];
            }
            const int32_t *s1p = p.rev_correction ? s1.p : nullptr;
            hipLaunchKernelGGL(gate_kernel, grid_for(n), dim3(256), 0, s, n, Lsq, s0.p, s1p, d_ms.p, qbegin, gflag.p);
            scan_u32(*this, tmp, gflag.p, gpos.p, n, false);
            const uint32_t n2 = scan_total(*this, gflag.p, gpos.p, n);
            if (n2) {
                q2.reserve(n2); t2.reserve(n2); qe2.reserve(n2); te2.reserve(n2); link.reserve(n2);
                s2.reserve(n2); q2o.reserve(n2); t2o.reserve(n2); eflag.reserve(n2); epos.reserve(n2);
                hipLaunchKernelGGL(gate_scatter_kernel, grid_for(n), dim3(256), 0, s, n, gflag.p, gpos.p, Lsq, Lst, qe0.p, te0.p,
                                   q2.p, t2.p, qe2.p, te2.p, link.p);
                if (p.sw_pk || dedup) fix_ambiguous_ends(*this, n2, q2.p, t2.p, qe2.p, te2.p, link.p, s0.p, qe0.p, te0.p, work, tmp, qb - qa);
            }
            hipLaunchKernelGGL(aln_basic_kernel, grid_for(n), dim3(256), 0, s, n, Lidx, s0.p, s1p, qe0.p, te0.p, gflag.p, alns_b);
            if (n2) {
                // start pass; results end up in the natural order of the gate-passer list (iota2 stands for the plan order)
                iota2.reserve(n2); s2s.reserve(n2); q2os.reserve(n2); t2os.reserve(n2);
                hipLaunchKernelGGL(iota_kernel, grid_for(n2), dim3(256), 0, s, n2, iota2.p);
                UC_HIP(hipMemsetAsync(d_cells.p, 0, 8, s));
                hipLaunchKernelGGL(cells_box_kernel, grid_for(n2), dim3(256), 0, s, n2, qe2.p, te2.p, d_cells.p);
                if (dedup && tab == 1) {
                    smkeep.reserve(n2); smpos.reserve(n2); partner.reserve(n2); uniq.reserve(n2);
                    UC_HIP(hipMemsetAsync(uniq.p, 0, (size_t)n2 * 4, s));
                    hipLaunchKernelGGL(sm_flag_kernel, grid_for(n2), dim3(256), 0, s, n2, link.p, mirror.p, gflag.p, gpos.p, qe2.p, te2.p,
                                       qe0.p, te0.p, smkeep.p, partner.p);
                    scan_u32(*this, tmp, smkeep.p, smpos.p, n2, false);
                    const uint32_t n2a = scan_total(*this, smkeep.p, smpos.p, n2);
                    q2a.reserve(n2a); t2a.reserve(n2a); qe2a.reserve(n2a); te2a.reserve(n2a); mapa.reserve(n2a);
                    hipLaunchKernelGGL(sm_gather_kernel, grid_for(n2), dim3(256), 0, s, n2, smkeep.p, smpos.p, q2.p, t2.p, qe2.p, te2.p,
                                       q2a.p, t2a.p, qe2a.p, te2a.p, mapa.p);
                    // round 1: known-score start pass (packed MODE 6; the optimum of the start pass is the forward score) - exact
                    // in one pass, and it reports whether a single row holds every optimal cell
                    sknown.reserve(n2a);
                    hipLaunchKernelGGL(sm_score_kernel, grid_for(n2a), dim3(256), 0, s, n2a, mapa.p, link.p, s0.p, sknown.p);
                    build_plan(*this, P2, tmp, n2a, q2a.p, t2a.p, qe2a.p, te2a.p, tab, nullptr, nullptr, sknown.p);
                    run_plan(*this, P2, 6, s2s.p, q2os.p, t2os.p, work, tmp);
                    hipLaunchKernelGGL(sm_scatter_kernel, grid_for(n2a), dim3(256), 0, s, n2a, P2.idx.p, mapa.p, s2s.p, q2os.p, t2os.p,
                                       s2.p, q2o.p, t2o.p, uniq.p);
                    hipLaunchKernelGGL(sm_resolve_kernel, grid_for(n2), dim3(256), 0, s, n2, partner.p, uniq.p, s2.p, q2o.p, t2o.p);
                    // round 2: the mirrors that could not take their partner's result
                    hipLaunchKernelGGL(amb_flag_kernel, grid_for(n2), dim3(256), 0, s, n2, q2o.p, smkeep.p);
                    scan_u32(*this, tmp, smkeep.p, smpos.p, n2, false);
                    const uint32_t n2b = scan_total(*this, smkeep.p, smpos.p, n2);
                    if (n2b) {
                        q2a.reserve(n2b); t2a.reserve(n2b); qe2a.reserve(n2b); te2a.reserve(n2b); mapa.reserve(n2b);
                        hipLaunchKernelGGL(sm_gather_kernel, grid_for(n2), dim3(256), 0, s, n2, smkeep.p, smpos.p, q2.p, t2.p, qe2.p, te2.p,
                                           q2a.p, t2a.p, qe2a.p, te2a.p, mapa.p);
                        sknown.reserve(n2b);
                        hipLaunchKernelGGL(sm_score_kernel, grid_for(n2b), dim3(256), 0, s, n2b, mapa.p, link.p, s0.p, sknown.p);
                        build_plan(*this, P2b, tmp, n2b, q2a.p, t2a.p, qe2a.p, te2a.p, pk_known_tab(n2b, qb - qa), nullptr, nullptr, sknown.p);
                        run_plan(*this, P2b, 6, s2s.p, q2os.p, t2os.p, work, tmp);   // known-score start pass (packed MODE 6)
                        hipLaunchKernelGGL(sm_scatter_kernel, grid_for(n2b), dim3(256), 0, s, n2b, P2b.idx.p, mapa.p, s2s.p, q2os.p, t2os.p,
                                           s2.p, q2o.p, t2o.p, (uint32_t *)nullptr);
                        stats.n_pk_reruns += n2b;
                    }
                } else {
                    build_plan(*this, P2, tmp, n2, q2.p, t2.p, qe2.p, te2.p, tab);
                    run_plan(*this, P2, 2, s2s.p, q2os.p, t2os.p, work, tmp);
                    hipLaunchKernelGGL(sm_scatter_kernel, grid_for(n2), dim3(256), 0, s, n2, P2.idx.p, (const uint32_t *)nullptr, s2s.p, q2os.p,
                                       t2os.p, s2.p, q2o.p, t2o.p, (uint32_t *)nullptr);
                }
                {
                    unsigned long long hc = 0;
                    UC_HIP(hipMemcpyAsync(&hc, d_cells.p, 8, hipMemcpyDeviceToHost, s));
                    UC_HIP(hipStreamSynchronize(s));
                    stats.cells_start += hc;
                }
                hipLaunchKernelGGL(finalize_kernel, grid_for(n2), dim3(256), 0, s, n2, iota2.p, link.p, Lidx, q2.p, t2.p, s2.p,
                                   q2o.p, t2o.p, ddb.len, p.cov, p.cov_mode, alns_b, eflag.p, mism.p);
                scan_u32(*this, tmp, eflag.p, epos.p, n2, false);
                uint32_t ne = scan_total(*this, eflag.p, epos.p, n2);
                if ((p.min_seq_id > 0.0f || p.want_tb) && ne) {
                    // sequence-identity gate / BLAST-tab statistics: (alignment length, identities[, gaps]) of the traceback on
                    // the box, computed by the MODE 3 pass of the int32 kernel for the pairs that passed the coverage gate
                    DevBuf<uint32_t> &q3 = A.tb_q3, &t3 = A.tb_t3, &src3 = A.tb_src3, &trun = A.tb_trun, &tpos = A.tb_tpos, &tpart = A.tb_tpart,
                                     &ttie = A.tb_ttie, &tlo = A.tb_tlo, &thi = A.tb_thi, &tmiss = A.tb_tmiss, &tchunk = A.tb_tchunk, &tcpos = A.tb_tcpos;
                    DevBuf<unsigned long long> &tbsize = A.tb_tbsize, &tboff = A.tb_tboff;
                    DevBuf<uint8_t> &tbm = A.tb_tbm;
                    DevBuf<int32_t> &qs3 = A.tb_qs3, &qe3 = A.tb_qe3, &ts3 = A.tb_ts3, &te3 = A.tb_te3, &pack3 = A.tb_pack3, &gaps3 = A.tb_gaps3, &sc3 = A.tb_sc3;
                    SwPlan &P3 = A.tb_P3;
                    // the plans of the passes behind us are dead (their results have been scattered): at sizes where the matrix budget is what is left of the
                    // device (nominal configs[3]: ~35 GB of plan arrays beside ~57 GiB of matrices per batch) they go back before the matrices are sized
                    // (r06, ADVICE r05: only when memory IS short, i.e. when what is free would not give the matrix buffer its full budget (64 GiB = 55 % of 116 GiB) -
                    // re-allocating ~35 GB of plan arrays in the next 256 M-pair batch costs 1-2 s, more than the few batches the room buys: UC_ALLOC_LOG at nominal
                    // configs[3] showed 443 GiB of 1-2 GiB blocks, 10-12 s per call, profiles/r06/c4_nominal_alloc.txt)
                    size_t free_now = 0, total_now = 0;
                    const bool mem_short = hipMemGetInfo(&free_now, &total_now) != hipSuccess || free_now + A.tb_tbm.cap < (120ull << 30);
                    if (n2 >= (8u << 20) && mem_short) {
                        UC_HIP(hipStreamSynchronize(s));
                        P1.release(); P2.release(); P2b.release(); A.rr_P2.release(); A.amb_P3.release();
                        if (dedup) P0.release();                   // (without sharing, Lsq / Lst / Lidx ARE P0's arrays)
                    }
                    trun.reserve(n2); tpos.reserve(n2); tpart.reserve(n2); ttie.reserve(n2);
                    UC_HIP(hipMemsetAsync(ttie.p, 0, (size_t)n2 * 4, s));
                    {
                        unsigned long long hc = 0;
                        UC_HIP(hipMemsetAsync(d_cells.p, 0, 8, s));
                        hipLaunchKernelGGL(cells_tb_kernel, grid_for(n2), dim3(256), 0, s, n2, eflag.p, link.p, Lidx, alns_b, d_cells.p);
                        UC_HIP(hipMemcpyAsync(&hc, d_cells.p, 8, hipMemcpyDeviceToHost, s));
                        UC_HIP(hipStreamSynchronize(s));
                        stats.cells_tb += hc;
                    }
                    // the flagged entries: gather, plan, run, apply.  pk: packed MODE 7 (H bytes of the box's diagonal band, tb_band_of) + walk
                    // kernel; otherwise the int32 kernel carries the statistics through the DP (MODE 3, one pass per statistic).
                    //
                    // r05, pk: ONE plan for all flagged pairs of the call, and the byte matrices are produced in batches cut ALONG that plan - a batch is
                    // a range of its class-sorted task list that fits the matrix budget, i.e. thousands of tasks of one or two length classes per
                    // launch.  (r04 cut the flagged LIST into batches and planned each: every batch then ran all 28 class kernels with a handful of
                    // tasks each - 442 batches at nominal configs[3], MODE 7 at 1.5 T cells/s against 5 T for the other passes:
                    // profiles/r05/sw_pass_timing_*.txt.)  band > 0: the walk marks the pairs whose path left the band in tmiss.
                    auto tb_batch = [&](const uint32_t *flag, bool pk, int band) {
                        scan_u32(*this, tmp, flag, tpos.p, n2, false);
                        const uint32_t nt = scan_total(*this, flag, tpos.p, n2);
                        if (!nt) return;
                        q3.reserve(nt); t3.reserve(nt); src3.reserve(nt); qs3.reserve(nt); qe3.reserve(nt); ts3.reserve(nt); te3.reserve(nt);
                        pack3.reserve(nt); gaps3.reserve(nt); sc3.reserve(nt);
                        hipLaunchKernelGGL(tb_gather_kernel, grid_for(n2), dim3(256), 0, s, n2, flag, tpos.p, iota2.p, link.p, Lidx, q2.p,
                                           t2.p, alns_b, q3.p, t3.p, qs3.p, qe3.p, ts3.p, te3.p, src3.p, sc3.p);
                        static const uint32_t tb_gaps[4] = {0u, 0u, 1u, 0u};
                        uint64_t launches = 0;
                        int passes = 1;
                        const bool timing = getenv("UC_TIMING") != nullptr;
                        if (!pk) {
                            build_plan(*this, P3, tmp, nt, q3.p, t3.p, qe3.p, te3.p, 2, qs3.p, ts3.p);
                            timed_ms_begin();
                            launches = launch_plan(*this, P3, 3, pack3.p, nullptr, nullptr, work);
                            if (p.want_tb) {   // second statistic of the same traceback: number of gaps (BLAST-tab "gapopen")
                                launches += launch_plan(*this, P3, 3, gaps3.p, nullptr, nullptr, work, tb_gaps);
                                passes = 2;
                            }
                            stats.sw_kernel_ms += timed_ms_end();
                        } else {
                            build_plan(*this, P3, tmp, nt, q3.p, t3.p, qe3.p, te3.p, 1, qs3.p, ts3.p, sc3.p, /*lpt=*/false);
                            const uint32_t n_pk3 = P3.pair_base[h_tab[1].n];      // every systolic class of table 1 is packed
                            const uint32_t nt_pk = P3.task_base[h_tab[1].n];      // ... and their tasks come first, in pair order
                            double tb_ms = 0;
                            unsigned long long total = 0;
                            uint32_t nbatch = 0;
                            if (P3.n > n_pk3) {                                   // long queries: int32 MODE 3 (no matrices)
                                timed_ms_begin();
                                launches += launch_plan(*this, P3, 7, pack3.p, nullptr, nullptr, work, nullptr, /*only_long=*/true);
                                if (p.want_tb) launches += launch_plan(*this, P3, 3, gaps3.p, nullptr, nullptr, work, tb_gaps, /*only_long=*/true);
                                tb_ms += timed_ms_end();
                            }
                            if (n_pk3) {
                                tbsize.reserve((size_t)n_pk3 + 1); tboff.reserve((size_t)n_pk3 + 1);
                                hipLaunchKernelGGL(tb_size_kernel, grid_for(n_pk3), dim3(256), 0, s, n_pk3, P3.key2.p, P3.segstart.p, P3.ste.p,
                                                   P3.sts.p, P3.sqs.p, P3.sqe.p, band, 1, tbsize.p);
                                size_t tbb = 0;
                                UC_HIP(rocprim::exclusive_scan(nullptr, tbb, tbsize.p, tboff.p, 0ull, (size_t)n_pk3, rocprim::plus<unsigned long long>(), s));
                                tmp.reserve(tbb + 256);
                                UC_HIP(rocprim::exclusive_scan(tmp.p, tbb, tbsize.p, tboff.p, 0ull, (size_t)n_pk3, rocprim::plus<unsigned long long>(), s));
                                // where every task's matrices begin, and its first pair: the host cuts the batches at task boundaries
                                DevBuf<unsigned long long> &toff = A.tb_toff;
                                toff.reserve((size_t)nt_pk + 1);
                                hipLaunchKernelGGL(tb_taskoff_kernel, grid_for(nt_pk), dim3(256), 0, s, nt_pk, P3.tasks.p, tboff.p, toff.p);
                                std::vector<unsigned long long> h_toff((size_t)nt_pk + 1);
                                std::vector<SwTask> h_task(nt_pk);
                                unsigned long long lo_ = 0, ls_ = 0;
                                UC_HIP(hipMemcpyAsync(h_toff.data(), toff.p, (size_t)nt_pk * 8, hipMemcpyDeviceToHost, s));
                                UC_HIP(hipMemcpyAsync(h_task.data(), P3.tasks.p, (size_t)nt_pk * sizeof(SwTask), hipMemcpyDeviceToHost, s));
                                UC_HIP(hipMemcpyAsync(&lo_, tboff.p + (n_pk3 - 1), 8, hipMemcpyDeviceToHost, s));
                                UC_HIP(hipMemcpyAsync(&ls_, tbsize.p + (n_pk3 - 1), 8, hipMemcpyDeviceToHost, s));
                                UC_HIP(hipStreamSynchronize(s));
                                total = lo_ + ls_;
                                h_toff[nt_pk] = total;
                                // Fresh device memory is not free: 25-60 ms per GiB on a box whose memory has been used before (tools/ubench/alloc_cost.hip,
                                // profiles/r04/alloc_*.log), so a call whose matrices would fit a few times over takes a third of them at a time, at least
                                // 16 GiB (about the buffer r04 took for an eighth of its whole-box matrices).  What the buffer already holds is free to use.
                                unsigned long long budget = tb_budget_bytes(tbm.cap);
                                if (!getenv("UC_TB_BUDGET_MB")) {
                                    const unsigned long long want = std::max<unsigned long long>(16ull << 30, total / 3);
                                    budget = std::min(budget, std::max<unsigned long long>(want, (unsigned long long)tbm.cap));
                                    // a buffer of a useful size that is there already IS the budget: a segment whose batches would be a little larger than
                                    // the last one's must not take a new block (r06)
                                    if (tbm.cap >= ((size_t)8 << 30)) budget = (unsigned long long)tbm.cap - 64;
                                }
                                A.tb_tbm_live = true;
                                struct Live { bool &f; ~Live() { f = false; } } live{A.tb_tbm_live};
                                // the batches first, then ONE buffer for the largest of them (r06: every batch that was a little larger than the buffer the first
                                // one had sized took a new ~30 GiB block - 12 allocations of 0.8 s each in a 14 s call at 500 proteomes on a box with untouched memory)
                                std::vector<std::pair<uint32_t, uint32_t>> batches;
                                unsigned long long max_bytes = 0;
                                for (uint32_t t0 = 0; t0 < nt_pk;) {
                                    uint32_t t1 = t0 + 1;                      // (a single task always runs, whatever the budget: tests force tiny budgets)
                                    while (t1 < nt_pk && h_toff[t1 + 1] - h_toff[t0] <= budget) t1++;
                                    batches.emplace_back(t0, t1);
                                    max_bytes = std::max(max_bytes, h_toff[t1] - h_toff[t0]);
                                    t0 = t1;
                                }
                                tbm.reserve_exact(std::max<unsigned long long>(max_bytes, std::min<unsigned long long>(total, budget)) + 64);
                                for (const auto &bt : batches) {
                                    const uint32_t t0 = bt.first, t1 = bt.second;
                                    const unsigned long long base = h_toff[t0];
                                    const uint32_t p0 = h_task[t0].begin, p1 = t1 < nt_pk ? h_task[t1].begin : n_pk3;
                                    P3.tbm = tbm.p - base; P3.tboff = tboff.p; P3.tb_band = band;      // (a pair's offset counts from the start of the WHOLE plan)
                                    timed_ms_begin();
                                    launches += launch_plan(*this, P3, 7, pack3.p, nullptr, nullptr, work, nullptr, false, t0, t1, /*skip_long=*/true);
                                    hipLaunchKernelGGL(tb_walk_kernel, grid_for(p1 - p0), dim3(256), 0, s, p0, p1, ddb, P3.key2.p, P3.st.p, P3.sqs.p, P3.sqe.p,
                                                       P3.sts.p, P3.ste.p, P3.saux.p, p.gap_open, p.gap_ext, band, 1, tbm.p - base, tboff.p, pack3.p, gaps3.p);
                                    tb_ms += timed_ms_end();
                                    nbatch++;
                                }
                                P3.tbm = nullptr; P3.tboff = nullptr; P3.tb_band = 0;
                            }
                            stats.sw_kernel_ms += tb_ms;
                            if (timing) fprintf(stderr, "unicore-cluster[timing]: sw pass mode 7: %u pairs, %llu cells, %.2f ms (%llu launches), %llu matrix bytes, band %d, %u batch(es)\n",
                                                P3.n, (unsigned long long)P3.cells, tb_ms, (unsigned long long)launches, total, band, nbatch);
                        }
                        stats.sw_kernel_launches += launches;
                        stats.sw_algorithmic_bytes += passes * P3.alg_bytes;
                        stats.cells_run += passes * P3.cells;
                        stats.n_sw_runs += (uint64_t)passes * P3.n;
                        hipLaunchKernelGGL(tb_apply_kernel, grid_for(nt), dim3(256), 0, s, nt, P3.idx.p, src3.p, pack3.p,
                                           p.want_tb ? gaps3.p : (const int32_t *)nullptr, iota2.p, link.p, Lidx, p.min_seq_id, alns_b,
                                           eflag.p, ttie.p, (pk && band > 0) ? tmiss.p : (uint32_t *)nullptr);
                    };
                    // the walk over H bytes needs neighbouring cells within 127 of each other: largest substitution score + gap open
                    // (uc_align.hip tb_walk_kernel); anything else takes the int32 pass
                    bool tb_bytes_ok = p.gap_open >= p.gap_ext && p.gap_ext >= 0;
                    {
                        int m3 = -128, ma = -128, n3 = 127, na = 127;
                        for (int k = 0; k < 21 * 21; k++) {
                            m3 = std::max<int>(m3, p.S3[k]); ma = std::max<int>(ma, p.SA[k]);
                            n3 = std::min<int>(n3, p.S3[k]); na = std::min<int>(na, p.SA[k]);
                        }
                        tb_bytes_ok = tb_bytes_ok && m3 + ma + p.gap_open <= 127 && n3 + na >= -127;
                    }
                    // r05: MODE 7 stores a diagonal band of the box, not the box (tb_band_of); the few pairs whose traceback leaves the band are redone
                    // with everything stored.  UC_TB_BAND = half-width W (default 48; 0 = always the whole box; the tests force 1 and 4)
                    const int tb_band_w = getenv("UC_TB_BAND") ? std::max(0, atoi(getenv("UC_TB_BAND"))) : 48;
                    // the plan of a MODE 7 call costs ~110 B per pair beside the matrices: the flagged list is worked off in segments of 32 M pairs (nominal
                    // configs[3]: 170 M flagged pairs per 256 M-pair batch of the stage), each planned and cut into matrix batches on its own; a segment
                    // still holds ~1.5 TB of matrices, i.e. dozens of one-class batches.  (Tests: a forced matrix budget also forces 1000-pair segments.)
                    auto tb_pk = [&](const uint32_t *flag, int band) {
                        const uint32_t seg = getenv("UC_TB_BUDGET_MB") ? 1000u : (32u << 20);
                        scan_u32(*this, tmp, flag, tcpos.p, n2, false);
                        const uint32_t nlo = scan_total(*this, flag, tcpos.p, n2);
                        if (nlo <= seg) { if (nlo) tb_batch(flag, true, band); return; }
                        for (uint32_t lo = 0; lo < nlo; lo += seg) {
                            hipLaunchKernelGGL(tb_chunk_kernel, grid_for(n2), dim3(256), 0, s, n2, flag, tcpos.p, lo, std::min<uint32_t>(nlo, lo + seg), tchunk.p);
                            tb_batch(tchunk.p, true, band);
                        }
                    };
                    auto run_tb = [&](const uint32_t *flag) {
                        if (!p.sw_pk || !tb_bytes_ok) { tb_batch(flag, false, 0); return; }
                        tlo.reserve(n2); thi.reserve(n2); tmiss.reserve(n2); tchunk.reserve(n2); tcpos.reserve(n2);
                        hipLaunchKernelGGL(tb_split_kernel, grid_for(n2), dim3(256), 0, s, n2, flag, link.p, Lidx, alns_b, SW_PK_OVF_HOST, tlo.p, thi.p);
                        tb_batch(thi.p, false, 0);                            // scores beyond the packed range: int32 MODE 3
                        if (tb_band_w > 0) UC_HIP(hipMemsetAsync(tmiss.p, 0, (size_t)n2 * 4, s));
                        tb_pk(tlo.p, tb_band_w);
                        if (tb_band_w > 0) tb_pk(tmiss.p, 0);                 // tracebacks that left their band: the whole box
                    };
                    if (dedup) {
                        hipLaunchKernelGGL(tbm_flag_kernel, grid_for(n2), dim3(256), 0, s, n2, eflag.p, link.p, Lidx, mirror.p, gflag.p, gpos.p,
                                           alns_b, trun.p, tpart.p);
                        run_tb(trun.p);
                        hipLaunchKernelGGL(tbm_resolve_kernel, grid_for(n2), dim3(256), 0, s, n2, tpart.p, ttie.p, link.p, Lidx, p.min_seq_id,
                                           alns_b, eflag.p, trun.p);
                        run_tb(trun.p);   // mirrors whose representative had a gap-direction tie on its traceback
                    } else {
                        UC_HIP(hipMemcpyAsync(trun.p, eflag.p, (size_t)n2 * 4, hipMemcpyDeviceToDevice, s));
                        run_tb(trun.p);
                    }
                    scan_u32(*this, tmp, eflag.p, epos.p, n2, false);
                    ne = scan_total(*this, eflag.p, epos.p, n2);
                }
                if (ne) {
                    d_e.reserve(2 * (size_t)ne);
                    hipLaunchKernelGGL(edge_scatter_kernel, grid_for(n2), dim3(256), 0, s, n2, eflag.p, epos.p, q2.p, t2.p, d_e.p);
                    if (edges_on_host && !edges.empty()) {   // a host list from an earlier call that was read back: continue on the device
                        d_edges.reserve(edges.size());
                        UC_HIP(hipMemcpyAsync(d_edges.p, edges.data(), edges.size() * 4, hipMemcpyHostToDevice, s));
                        n_edges_dev = edges.size() / 2;
                    }
                    d_edges.grow_preserve(2 * (n_edges_dev + ne), 2 * n_edges_dev, s);
                    UC_HIP(hipMemcpyAsync(d_edges.p + 2 * n_edges_dev, d_e.p, 2 * (size_t)ne * 4, hipMemcpyDeviceToDevice, s));
                    n_edges_dev += ne;
                    edges_on_host = false;
                }
            }
            uint32_t bad = 0;
            UC_HIP(hipMemcpyAsync(&bad, mism.p, 4, hipMemcpyDeviceToHost, s));
            UC_HIP(hipStreamSynchronize(s));
            UC_HIP(hipGetLastError());
            if (bad) fail(UC_ERR_GENERIC, "start pass score differs from the forward score for %u pairs", bad);
            stats.n_gapped_alignments += n;
            stats.n_start_alignments += n2;
            if (alns_b != d_alns.p + b) {
                hipLaunchKernelGGL(len_gate_putback_kernel, grid_for(n), dim3(256), 0, s, n, A.lg_map.p, (const uc_aln *)alns_b, d_alns.p + b);
                UC_HIP(hipStreamSynchronize(s));
            }
        }
        qa = qb;
    }
    alns_valid = true;
    last_align_hits = qbegin == 0 && qend == hdb.n ? n_hits : 0;      // what the buffers of this set have just served (uc_prefilter.hip: scratch_release_target)
    stats.n_edges = edges_on_host ? edges.size() / 2 : n_edges_dev;
    stats.algorithmic_bytes[UC_ST_GAPPED] = stats.sw_algorithmic_bytes;
    stats.stage_seconds[UC_ST_GAPPED] += tm.seconds();
}

// ---- E7 graph construction on the device (the greedy cover follows below; uc_setcover.cpp is the all-host variant behind uc_setcover) ----
// both directions of every accepted pair as 64-bit keys, self loops -> all-ones (sorted last, dropped)
__global__ void __launch_bounds__(256) edge_key_kernel(uint64_t n_edges, const uint32_t *e, uint32_t n, uint64_t *key, uint32_t *bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_edges; i += (uint64_t)gridDim.x * 256) {
        const uint32_t a = e[2 * i], b = e[2 * i + 1];
        if (a >= n || b >= n) { atomicAdd(bad, 1u); key[2 * i] = ~0ull; key[2 * i + 1] = ~0ull; continue; }
        key[2 * i] = a == b ? ~0ull : ((uint64_t)a << 32) | b;
        key[2 * i + 1] = a == b ? ~0ull : ((uint64_t)b << 32) | a;
    }
}
__global__ void __launch_bounds__(256) edge_uniq_kernel(uint64_t m, const uint64_t *key, uint32_t *flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (uint64_t)gridDim.x * 256)
        flag[i] = (key[i] != ~0ull && (i == 0 || key[i] != key[i - 1])) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) edge_adj_kernel(uint64_t m, const uint64_t *key, const uint32_t *flag, const uint32_t *pos,
                                                       uint64_t *ukey, uint32_t *adj) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (uint64_t)gridDim.x * 256)
        if (flag[i]) { ukey[pos[i]] = key[i]; adj[pos[i]] = (uint32_t)key[i]; }
}
__global__ void __launch_bounds__(256) edge_off_kernel(uint32_t n, const uint64_t *ukey, uint64_t mu, uint64_t *off) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i <= n; i += gridDim.x * 256) {
        const uint64_t want = (uint64_t)i << 32;
        uint64_t lo = 0, hi = mu;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (ukey[mid] < want) lo = mid + 1; else hi = mid; }
        off[i] = lo;
    }
}

const std::vector<uint32_t> &Engine::host_edges() {
    if (!edges_on_host) {
        UC_HIP(hipSetDevice(device));
        edges.resize(2 * n_edges_dev);
        if (n_edges_dev) UC_HIP(hipMemcpy(edges.data(), d_edges.p, 2 * n_edges_dev * 4, hipMemcpyDeviceToHost));
        edges_on_host = true;
    }
    return edges;
}

// ---- E7 on the device: the SAME greedy cover as uc_setcover.cpp (most unassigned nodes covered first, ties: smallest id), in parallel rounds ----
// The sequential rule picks the unassigned node u with the largest key (count, -id), count = 1 + unassigned neighbours.  A pick changes counts only
// within two hops THROUGH UNASSIGNED nodes (it assigns its unassigned neighbours v, which lowers the count of their unassigned neighbours w), and counts
// only ever fall.  So a node whose key is the maximum of its two-hop neighbourhood through unassigned nodes will be picked by the sequential rule with
// exactly its present count before anything near it changes: every such local maximum can be picked AT ONCE (no two of them share an unassigned
// neighbour: they would be within two hops of each other).  Rounds of (1) m1[v] = max key over the unassigned closed neighbourhood of v, (2) u is picked
// iff key(u) = max m1 over its unassigned closed neighbourhood, (3) picks take their unassigned neighbours, (4) the counts of the unassigned neighbours
// of everything newly assigned drop.  Identical assignment to the host cover (tests/test_gpu_parity.py, random graphs included); a wave per node.
constexpr uint32_t SC_NONE = 0xFFFFFFFFu;
__device__ __forceinline__ uint64_t sc_key(uint32_t w, const uint32_t *assign, const uint32_t *cnt) {
    return assign[w] == SC_NONE ? ((uint64_t)cnt[w] << 32) | (uint64_t)(0xFFFFFFFFu - w) : 0ull;
}
__device__ __forceinline__ uint64_t sc_wave_max(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) { const uint64_t x = __shfl_xor(v, o, 64); v = x > v ? x : v; }
    return v;
}
__global__ void __launch_bounds__(256) sc_init_kernel(uint32_t n, const uint64_t *off, uint32_t *assign, uint32_t *cnt, uint32_t *work) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        assign[i] = SC_NONE; cnt[i] = (uint32_t)(off[i + 1] - off[i]) + 1u; work[i] = i;
    }
}
// ctr: [0] nodes in `work`, [1] picks, [2] newly assigned, [3] nodes in the next work list
__global__ void __launch_bounds__(256) sc_m1_kernel(const uint32_t *ctr, const uint32_t *work, const uint64_t *off, const uint32_t *adj, const uint32_t *assign,
                                                    const uint32_t *cnt, uint64_t *m1) {
    const uint32_t nw = ctr[0], lane = threadIdx.x & 63;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < nw; i += gridDim.x * 4) {
        const uint32_t v = work[i];
        uint64_t best = 0;
        for (uint64_t e = off[v] + lane; e < off[v + 1]; e += 64) { const uint64_t k = sc_key(adj[e], assign, cnt); best = k > best ? k : best; }
        best = sc_wave_max(best);
        const uint64_t own = sc_key(v, assign, cnt);
        if (lane == 0) m1[v] = own > best ? own : best;
    }
}
__global__ void __launch_bounds__(256) sc_pick_kernel(uint32_t *ctr, const uint32_t *work, const uint64_t *off, const uint32_t *adj, const uint32_t *assign,
                                                      const uint32_t *cnt, const uint64_t *m1, uint32_t *picks) {
    const uint32_t nw = ctr[0], lane = threadIdx.x & 63;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < nw; i += gridDim.x * 4) {
        const uint32_t u = work[i];
        uint64_t best = 0;
        for (uint64_t e = off[u] + lane; e < off[u + 1]; e += 64) {
            const uint32_t w = adj[e];
            const uint64_t k = assign[w] == SC_NONE ? m1[w] : 0ull;
            best = k > best ? k : best;
        }
        best = sc_wave_max(best);
        const uint64_t mine = m1[u];
        best = mine > best ? mine : best;
        if (lane == 0 && best == sc_key(u, assign, cnt)) picks[atomicAdd(&ctr[1], 1u)] = u;
    }
}
__global__ void __launch_bounds__(256) sc_apply_kernel(uint32_t *ctr, const uint32_t *picks, const uint64_t *off, const uint32_t *adj, uint32_t *assign, uint32_t *newly) {
    const uint32_t np = ctr[1], lane = threadIdx.x & 63;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < np; i += gridDim.x * 4) {
        const uint32_t u = picks[i];
        if (lane == 0) { assign[u] = u; newly[atomicAdd(&ctr[2], 1u)] = u; }
        for (uint64_t e = off[u] + lane; e < off[u + 1]; e += 64) {
            const uint32_t w = adj[e];
            if (assign[w] == SC_NONE) { assign[w] = u; newly[atomicAdd(&ctr[2], 1u)] = w; }     // no other pick of this round can reach w
        }
    }
}
__global__ void __launch_bounds__(256) sc_dec_kernel(const uint32_t *ctr, const uint32_t *newly, const uint64_t *off, const uint32_t *adj, const uint32_t *assign, uint32_t *cnt) {
    const uint32_t nn = ctr[2], lane = threadIdx.x & 63;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < nn; i += gridDim.x * 4) {
        const uint32_t v = newly[i];
        for (uint64_t e = off[v] + lane; e < off[v + 1]; e += 64) {
            const uint32_t w = adj[e];
            if (assign[w] == SC_NONE) atomicSub(&cnt[w], 1u);
        }
    }
}
__global__ void __launch_bounds__(256) sc_compact_kernel(uint32_t *ctr, const uint32_t *work, const uint32_t *assign, uint32_t *next) {
    const uint32_t nw = ctr[0];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nw; i += gridDim.x * 256) {
        const uint32_t v = work[i];
        if (assign[v] == SC_NONE) next[atomicAdd(&ctr[3], 1u)] = v;
    }
}
// the edges among the still-unassigned nodes of the work list (v < w once each): out == nullptr counts them, else appends them behind an atomic cursor
__global__ void __launch_bounds__(256) sc_induced_kernel(uint32_t left, const uint32_t *work, const uint64_t *off, const uint32_t *adj, const uint32_t *assign,
                                                         uint32_t *out, unsigned long long *cursor) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < left; i += gridDim.x * 256) {
        const uint32_t v = work[i];
        if (assign[v] != SC_NONE) continue;
        uint32_t c = 0;
        for (uint64_t k = off[v]; k < off[(size_t)v + 1]; k++) { const uint32_t w = adj[k]; c += (w > v && assign[w] == SC_NONE) ? 1u : 0u; }
        if (!c) continue;
        const unsigned long long base = atomicAdd(cursor, (unsigned long long)c);
        if (!out) continue;
        unsigned long long o = base;
        for (uint64_t k = off[v]; k < off[(size_t)v + 1]; k++) {
            const uint32_t w = adj[k];
            if (w > v && assign[w] == SC_NONE) { out[2 * o] = v; out[2 * o + 1] = w; o++; }
        }
    }
}
__global__ void sc_next_round_kernel(uint32_t *ctr) { ctr[0] = ctr[3]; ctr[1] = 0; ctr[2] = 0; ctr[3] = 0; }

void Engine::set_cover_own_edges(uint32_t n, uint32_t *assign) {
    if (edges_on_host) { set_cover_device(n, edges.data(), edges.size() / 2, assign); return; }
    if (n_edges_dev >= (1ull << 31)) { const std::vector<uint32_t> &h = host_edges(); set_cover(n, h.data(), h.size() / 2, assign); return; }
    set_cover_graph(n, nullptr, d_edges.p, n_edges_dev, assign);
}

void Engine::set_cover_device(uint32_t n, const uint32_t *h_edges, uint64_t n_edges, uint32_t *assign) {
    if (n_edges >= (1ull << 31)) { set_cover(n, h_edges, n_edges, assign); return; }   // 32-bit scan positions below
    set_cover_graph(n, h_edges, nullptr, n_edges, assign);
}

// the graph (CSR, both directions, duplicates and self loops removed) on the device from a host or a device edge list, then the
// greedy cover on the device as well (parallel rounds of local maxima: the host cover's assignment, uc_setcover.cpp, exactly)
void Engine::set_cover_graph(uint32_t n, const uint32_t *h_edges, const uint32_t *dev_edges, uint64_t n_edges, uint32_t *assign) {
    PressureScope ps(*this, 1);
    UC_HIP(hipSetDevice(device));
    Timer t_graph;
    AlignScratch &A = scratch_of(*this);
    if (n_edges) {
        const uint64_t m = 2 * n_edges;
        DevBuf<uint32_t> &d_e = A.sc_e, &flag = A.sc_flag, &pos = A.sc_pos, &d_adj = A.sc_dadj, &bad = A.sc_bad;
        DevBuf<uint64_t> &key = A.sc_key, &key2 = A.sc_key2, &ukey = A.sc_ukey, &d_off = A.sc_doff;
        DevBuf<char> &tmp = A.sc_tmp;
        key.reserve(m); key2.reserve(m); flag.reserve(m); pos.reserve(m); bad.reserve(1); d_off.reserve((size_t)n + 1);
        const uint32_t *e_in = dev_edges;
        if (!e_in) {
            d_e.reserve(m);
            UC_HIP(hipMemcpyAsync(d_e.p, h_edges, m * 4, hipMemcpyHostToDevice, stream));
            e_in = d_e.p;
        }
        UC_HIP(hipMemsetAsync(bad.p, 0, 4, stream));
        hipLaunchKernelGGL(edge_key_kernel, grid_for(n_edges), dim3(256), 0, stream, n_edges, e_in, n, key.p, bad.p);
        size_t tb = 0;
        UC_HIP(rocprim::radix_sort_keys(nullptr, tb, key.p, key2.p, (size_t)m, 0u, 64u, stream));
        tmp.reserve(tb + 256);
        UC_HIP(rocprim::radix_sort_keys(tmp.p, tb, key.p, key2.p, (size_t)m, 0u, 64u, stream));
        hipLaunchKernelGGL(edge_uniq_kernel, grid_for(m), dim3(256), 0, stream, m, key2.p, flag.p);
        scan_u32(*this, tmp, flag.p, pos.p, (uint32_t)m, false);
        const uint32_t mu = scan_total(*this, flag.p, pos.p, (uint32_t)m);
        uint32_t hbad = 0;
        UC_HIP(hipMemcpy(&hbad, bad.p, 4, hipMemcpyDeviceToHost));
        if (hbad) fail(UC_ERR_ARGS, "set cover: %u edges with an endpoint out of range", hbad);
        ukey.reserve(std::max<uint32_t>(mu, 1)); d_adj.reserve(std::max<uint32_t>(mu, 1));
        hipLaunchKernelGGL(edge_adj_kernel, grid_for(m), dim3(256), 0, stream, m, key2.p, flag.p, pos.p, ukey.p, d_adj.p);
        hipLaunchKernelGGL(edge_off_kernel, grid_for((uint64_t)n + 1), dim3(256), 0, stream, n, ukey.p, (uint64_t)mu, d_off.p);
        UC_HIP(hipGetLastError());
        const bool timing = getenv("UC_TIMING") != nullptr;
        if (timing) { UC_HIP(hipStreamSynchronize(stream)); fprintf(stderr, "set_cover_device: graph on the GPU %.2f ms\n", t_graph.seconds() * 1e3); }
        // the greedy cover itself, on the device too (parallel rounds, see above): neither the adjacency (2 x edges x 4 bytes) nor the serial
        // host sweep — rank 0's Amdahl term of an N-GPU pass — is left; only the assignment comes back
        Timer t_greedy;
        DevBuf<uint32_t> &d_assign = A.sc_assign, &d_cnt = A.sc_cnt, &w0 = A.sc_work, &w1 = A.sc_work2, &picks = A.sc_picks, &newly = A.sc_newly, &ctr = A.sc_ctr;
        DevBuf<uint64_t> &m1 = A.sc_m1;
        d_assign.reserve(n); d_cnt.reserve(n); w0.reserve(n); w1.reserve(n); picks.reserve(n); newly.reserve(n); ctr.reserve(4); m1.reserve(n);
        hipLaunchKernelGGL(sc_init_kernel, grid_for(n), dim3(256), 0, stream, n, d_off.p, d_assign.p, d_cnt.p, w0.p);
        const uint32_t h_ctr[4] = {n, 0, 0, 0};
        UC_HIP(hipMemcpyAsync(ctr.p, h_ctr, 16, hipMemcpyHostToDevice, stream));
        uint32_t left = n, rounds = 0;
        bool host_tail = false;
        uint32_t *cur = w0.p, *nxt = w1.p;
        while (left) {
            const dim3 gw((uint32_t)std::min<uint64_t>(((uint64_t)left + 3) / 4, 1u << 16));
            hipLaunchKernelGGL(sc_m1_kernel, gw, dim3(256), 0, stream, (const uint32_t *)ctr.p, (const uint32_t *)cur, d_off.p, d_adj.p, d_assign.p, d_cnt.p, m1.p);
            hipLaunchKernelGGL(sc_pick_kernel, gw, dim3(256), 0, stream, ctr.p, (const uint32_t *)cur, d_off.p, d_adj.p, d_assign.p, d_cnt.p, m1.p, picks.p);
            hipLaunchKernelGGL(sc_apply_kernel, gw, dim3(256), 0, stream, ctr.p, (const uint32_t *)picks.p, d_off.p, d_adj.p, d_assign.p, newly.p);
            hipLaunchKernelGGL(sc_dec_kernel, gw, dim3(256), 0, stream, (const uint32_t *)ctr.p, (const uint32_t *)newly.p, d_off.p, d_adj.p, d_assign.p, d_cnt.p);
            hipLaunchKernelGGL(sc_compact_kernel, grid_for(left), dim3(256), 0, stream, ctr.p, (const uint32_t *)cur, d_assign.p, nxt);
            uint32_t h[4];
            UC_HIP(hipMemcpyAsync(h, ctr.p, 16, hipMemcpyDeviceToHost, stream));
            hipLaunchKernelGGL(sc_next_round_kernel, dim3(1), dim3(1), 0, stream, ctr.p);
            UC_HIP(hipStreamSynchronize(stream));
            if (h[1] == 0) fail(UC_ERR_GENERIC, "set cover: a round picked nothing with %u nodes left", left);    // cannot happen: the global maximum is always a local one
            left = h[3];
            std::swap(cur, nxt);
            rounds++;
            // Family graphs finish in a handful of rounds; a chain-like remainder (equal counts, ids ascending along a path) gives ONE pick per round -
            // n / 3 rounds of six launches and a host synchronisation each.  When the rounds stop paying, the rest - the greedy cover of
            // the subgraph the unassigned nodes induce, which is all the sequential rule still looks at - is finished by the host cover.
            if (left && rounds >= 32 && h[1] < 64) { host_tail = true; break; }
        }
        UC_HIP(hipMemcpyAsync(assign, d_assign.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipStreamSynchronize(stream));
        UC_HIP(hipGetLastError());
        if (host_tail) {
            // r06: only the subgraph the `left` unassigned nodes induce goes to the host - their ids and the edges among them, gathered on the
            // device (count pass, then an append pass; the order of the edge list does not matter to the cover) - not the whole adjacency (4 B x 10^9 entries
            // over PCIe plus an O(n) scan at nominal sizes, however few nodes were left)
            DevBuf<unsigned long long> ecnt;
            ecnt.reserve(1);
            UC_HIP(hipMemsetAsync(ecnt.p, 0, 8, stream));
            hipLaunchKernelGGL(sc_induced_kernel, grid_for(left), dim3(256), 0, stream, left, (const uint32_t *)cur, d_off.p, d_adj.p, d_assign.p, (uint32_t *)nullptr, ecnt.p);
            unsigned long long ne2 = 0;
            UC_HIP(hipMemcpyAsync(&ne2, ecnt.p, 8, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipStreamSynchronize(stream));
            DevBuf<uint32_t> de2;
            de2.reserve(2 * std::max<unsigned long long>(ne2, 1));
            UC_HIP(hipMemsetAsync(ecnt.p, 0, 8, stream));
            if (ne2) hipLaunchKernelGGL(sc_induced_kernel, grid_for(left), dim3(256), 0, stream, left, (const uint32_t *)cur, d_off.p, d_adj.p, d_assign.p, de2.p, ecnt.p);
            std::vector<uint32_t> back(left), e2(2 * (size_t)ne2);
            UC_HIP(hipMemcpyAsync(back.data(), cur, (size_t)left * 4, hipMemcpyDeviceToHost, stream));
            if (ne2) UC_HIP(hipMemcpyAsync(e2.data(), de2.p, 2 * (size_t)ne2 * 4, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipStreamSynchronize(stream));
            UC_HIP(hipGetLastError());
            std::sort(back.begin(), back.end());                                   // ids keep their order: so do the ties
            auto sub = [&](uint32_t v) { return (uint32_t)(std::lower_bound(back.begin(), back.end(), v) - back.begin()); };
            for (uint32_t &x : e2) x = sub(x);
            std::vector<uint32_t> a2(back.size());
            set_cover((uint32_t)back.size(), e2.data(), e2.size() / 2, a2.data());
            for (size_t i = 0; i < back.size(); i++) assign[back[i]] = back[a2[i]];
        }
        if (timing) fprintf(stderr, "set_cover_device: greedy cover on the GPU %.2f ms in %u rounds%s\n", t_greedy.seconds() * 1e3, rounds,
                            host_tail ? ", the chain-like rest on the host" : "");
        return;
    }
    for (uint32_t i = 0; i < n; i++) assign[i] = i;      // no edges: every node is its own representative
}

void preload_align_module() { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)plan_key_kernel); }
}  // namespace uc
