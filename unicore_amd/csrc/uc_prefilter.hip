// uc_prefilter.hip — stages E1-E4 on the GPU: k-mer index build, similar-k-mer matching with the
// same-diagonal double-hit rule, ungapped rescoring, per-query top-M selection.
// Stands for the MMseqs2-style prefilter inside `foldseek cluster`
// (call site /root/reference/src/modules/cluster.rs:45-56; algorithm SURVEY.md A.2; spec UC-1 E1-E4).
//
// All of this is HBM-bound gather/scan/sort work:
//   E1  extract (k-mer, seq, pos) per target residue (coalesced scan of the 3Di track) -> radix sort by
//       k-mer -> CSR offsets by binary search per k-mer slot (20^6 + 1 u32 = 256 MB, lives in HBM/MALL).
//       Large target ranges are indexed in chunks; the per-query lists of the chunks are merged on the device.
//   E2  per query residue: enumerate similar k-mers (sorted-letter DFS with score bound; tables in LDS) and keep
//       their non-empty index ranges ("runs", staged per wave in LDS) -> runs sorted by query position -> one
//       workgroup per query expands its runs twice: sweep 1 marks hash(target, diagonal) in two LDS bitmaps
//       (seen once / twice, two hash positions per key), sweep 2 keeps the keys seen twice -> compaction ->
//       radix sort of the surviving keys -> one pass run-length-counts diagonals per (query, target) and keeps
//       the best one (candidates appended per wave in blocks).  min_diag_hits 1 uses a plain expansion.
//   E3  ungapped diagonal score per candidate.   E4  sort by (query, score desc, target), rank inside the
//       query's run < max_seqs, scatter into the device-resident hit lists.
// Sort and scan primitives come from rocPRIM; every kernel below is hand-written for wave64.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include <rocprim/rocprim.hpp>

#include "uc_engine.h"

namespace uc {

constexpr uint32_t KMER_INVALID = 0xFFFFFFFFu;

struct KmerCfg {
    int koff[K];
    int span;
    int thr;
};

// sequence id containing padded residue offset p, searched in off[lo..hi)
__device__ __forceinline__ uint32_t find_seq(const uint32_t *off, uint32_t lo, uint32_t hi, uint32_t p) {
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------- E1: index build
__global__ void __launch_bounds__(256) kmer_extract_kernel(const DeviceDb db, uint32_t tbegin, uint32_t tend, KmerCfg cfg,
                                                           uint32_t p0, uint32_t p1, uint32_t *keys, uint64_t *vals,
                                                           uint32_t *vals32 /* compact entries instead of vals */, int pshift) {
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < p1 - p0; idx += (uint64_t)gridDim.x * 256) {
        const uint32_t p = p0 + (uint32_t)idx;
        const uint32_t s = find_seq(db.off, tbegin, tend, p);
        const uint32_t j = p - db.off[s], len = db.len[s];
        uint32_t key = KMER_INVALID;
        if (j + cfg.span <= len && j <= 65535u) {
            uint32_t v = 0, mul = 1;
            bool ok = true;
#pragma unroll
            for (int m = 0; m < K; m++) {
                const uint32_t c = db.s3[p + cfg.koff[m]];
                ok &= c < KA;
                v += c * mul;
                mul *= KA;
            }
            if (ok) key = v;
        }
        keys[idx] = key;
        if (vals32) vals32[idx] = ((s - tbegin) << pshift) | (j & ((1u << pshift) - 1u));
        else vals[idx] = ((uint64_t)s << 16) | j;
    }
}

__global__ void __launch_bounds__(256) kmer_offsets_kernel(const uint32_t *keys, uint32_t n, uint32_t *koff) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k <= KSPACE; k += (uint64_t)gridDim.x * 256) {
        uint32_t lo = 0, hi = n;   // first index with key >= k
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (keys[mid] < (uint32_t)k) lo = mid + 1; else hi = mid;
        }
        koff[k] = lo;
    }
}

// presence bitmap of the chunk's k-mers (bit v = the index holds k-mer v): 8 MB for the 20^6 k-mers — resident in every XCD's L2, where the
// 256 MB offset table is not.  At high sensitivity the target chunks are small (density cuts: ~5 M residues at 100 proteomes, < 10 % of the
// k-mer space occupied) and nine of ten similar k-mers miss; the bitmap answers those without touching the offset table (r4: prefilter
// kernels 7.67 -> 6.93 s at 50 proteomes, 26.0 -> 23.5 s at 100 proteomes with configs[3]'s options; nothing at -s 4, where most k-mers occur).
__global__ void __launch_bounds__(256) kmer_bits_kernel(const uint32_t *koff, uint32_t *bits) {
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < (KSPACE + 31) / 32; w += (uint64_t)gridDim.x * 256) {
        uint32_t m = 0;
        const uint64_t v0 = w * 32;
        uint32_t prev = koff[v0];
        for (int b = 0; b < 32 && v0 + b < KSPACE; b++) {
            const uint32_t nxt = koff[v0 + b + 1];
            m |= (nxt != prev ? 1u : 0u) << b;
            prev = nxt;
        }
        bits[w] = m;
    }
}

// ---------------------------------------------------------------- E2: similar k-mers
constexpr int SIM_NEED_MAX = 49;   // substitution scores lie in [-48, 48] (checked by the host): a needed score beyond either end selects all / no letters
struct SimTables {           // LDS: per query letter a, target letters sorted by score descending
    int8_t sc[KA][KA];
    uint8_t ord[KA][KA];
    int8_t rowmax[KA];
    uint8_t lcnt[KA][2 * SIM_NEED_MAX + 1];    // [a][need + SIM_NEED_MAX]: how many letters b have S(a, b) >= need (they are the first lcnt of the sorted row) ...
    uint32_t pmask[KA][KA + 1];                // ... and [a][n]: the first n letters of a's sorted row as a bit mask (r05: the DFS's last level in ONE step)
};

__device__ void build_sim_tables(SimTables &t, const int8_t *S3) {
    for (int a = threadIdx.x; a < KA; a += blockDim.x) {
        int8_t sc[KA];
        uint8_t ord[KA];
        for (int b = 0; b < KA; b++) { sc[b] = S3[a * 21 + b]; ord[b] = (uint8_t)b; }
        for (int i = 1; i < KA; i++) {   // insertion sort: score desc, letter asc
            const int8_t s = sc[i];
            const uint8_t o = ord[i];
            int k = i - 1;
            while (k >= 0 && (sc[k] < s)) { sc[k + 1] = sc[k]; ord[k + 1] = ord[k]; k--; }
            sc[k + 1] = s;
            ord[k + 1] = o;
        }
        for (int b = 0; b < KA; b++) { t.sc[a][b] = sc[b]; t.ord[a][b] = ord[b]; }
        t.rowmax[a] = sc[0];
    }
    for (int i = threadIdx.x; i < KA * (2 * SIM_NEED_MAX + 1); i += blockDim.x) {
        const int a = i / (2 * SIM_NEED_MAX + 1), need = i % (2 * SIM_NEED_MAX + 1) - SIM_NEED_MAX;
        int n = 0;
        for (int b = 0; b < KA; b++) n += S3[a * 21 + b] >= need ? 1 : 0;
        t.lcnt[a][i % (2 * SIM_NEED_MAX + 1)] = (uint8_t)n;
    }
    __syncthreads();                            // (ord is read below)
    for (int i = threadIdx.x; i < KA * (KA + 1); i += blockDim.x) {
        const int a = i / (KA + 1), n = i % (KA + 1);
        uint32_t m = 0;
        for (int j = 0; j < n; j++) m |= 1u << t.ord[a][j];
        t.pmask[a][n] = m;
    }
}

// shared by the count and the emit pass: decode the query position handled by this thread
__device__ __forceinline__ bool query_kmer_at(const DeviceDb &db, const KmerCfg &cfg, uint32_t qbegin, uint32_t qend,
                                              uint32_t p, uint32_t *q, uint32_t *i, uint32_t c[K]) {
    const uint32_t s = find_seq(db.off, qbegin, qend, p);
    const uint32_t j = p - db.off[s], len = db.len[s];
    *q = s;
    *i = j;
    if (!(j + cfg.span <= len && j <= 65535u)) return false;
    bool ok = true;
#pragma unroll
    for (int m = 0; m < K; m++) { c[m] = db.s3[p + cfg.koff[m]]; ok &= c[m] < KA; }
    return ok;
}

// ---- E2 pass 1: enumerate similar k-mers (sorted-letter DFS with score bound), keep the non-empty index ranges ("runs") ----
// A run is (first index entry, entry count, query position inside the batch); run order is irrelevant (keys get sorted).
// A lane per query position running its own nested loops leaves most lanes idle: the number of similar k-mers per
// position is heavy-tailed (mean ~20, maximum > 1000), a wave lasts as long as its worst position (~0.18 SIMD
// efficiency).  Here the DFS is an explicit state machine (level, index per level and the six letters packed into
// registers; score and value so far kept incrementally): every wave step advances every lane by ONE node, and a lane that
// finishes its position takes the next one of the wave's region at once, so all lanes stay busy until the region is
// exhausted — which pays once a lane works through many positions, hence one contiguous region per wave and query
// batches as large as the key buffer allows.  Positions are decoded 64 at a time into a small pool (the sequence search
// and letter loads are ~10 us of dependent loads that must not sit in front of every hand-out).  Leaves (k-mer value,
// position) are appended to a wave queue by ballot rank — no per-lane atomics under divergence — and the offset-table
// lookups are done 64 at a time when the queue fills, fully parallel and off the enumeration's critical path; the
// non-empty ranges are staged per wave and written out behind ONE global atomic per RUN_STAGE runs.
constexpr int RUN_STAGE = 256;      // staged runs per wave

struct RunList {
    uint32_t *pidx;     // query position inside the batch
    uint64_t *val;      // [ entry count : 32 | first index entry : 32 ]
    uint64_t cap;
};

// counters: [0] similar k-mers, [1] kept candidates, [2] ungapped overlap, [3] run cursor, [4] k-mer hits, [5] key cursor
constexpr int LEAFQ = 256;          // leaf queue entries per wave
constexpr int POSQ = 128;           // decoded positions per wave (ring)
constexpr int SIM_MIN_WAVE_POS = 256;   // fewer positions per wave than this: fewer workgroups
constexpr int SIM_MAX_BLOCKS = 1024;     // 256 CUs x 4 resident workgroups (37 KB of LDS each)

__global__ void __launch_bounds__(256) sim_runs_kernel(const DeviceDb db, KmerCfg cfg, uint32_t qbegin, uint32_t qend,
                                                            uint32_t p0, uint32_t p1, const uint32_t *koff, RunList out,
                                                            unsigned long long *counters,
                                                            const uint32_t *dk /* distinct mode: work item i = k-mer value dk[i] */,
                                                            uint32_t *nsim_k /* distinct mode: similar k-mers per work item */,
                                                            uint32_t item_stride /* distinct mode: every item_stride-th item only (run-count estimate) */,
                                                            const uint32_t *kbits = nullptr /* presence bitmap of the chunk's k-mers (nullable) */) {
    __shared__ SimTables tab;
    __shared__ uint32_t s_mul[K];
    __shared__ uint32_t s_n[4];
    __shared__ uint64_t s_val[4][RUN_STAGE];
    __shared__ uint32_t s_pi[4][RUN_STAGE];
    __shared__ uint32_t s_qv[4][LEAFQ], s_qp[4][LEAFQ], s_qm[4][LEAFQ];
    __shared__ uint32_t s_pc[4][POSQ], s_pp[4][POSQ];
    __shared__ uint64_t s_pr[4][POSQ];
    build_sim_tables(tab, db.S3);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // level L of the DFS decides k-mer position K - 1 - L, so the LAST level is the least significant digit of the k-mer value: the (up to 20) similar
    // k-mers under one five-letter prefix are consecutive values - one or two words of the presence bitmap, neighbouring entries of the offset table
    if (threadIdx.x < K) { uint32_t m = 1; for (int i = 0; i < K - 1 - (int)threadIdx.x; i++) m *= KA; s_mul[threadIdx.x] = m; }
    if (lane == 0) s_n[wv] = 0;
    __syncthreads();
    volatile uint32_t *vn = &s_n[wv];
    volatile uint64_t *vval = s_val[wv];
    volatile uint32_t *vpi = s_pi[wv];
    volatile uint32_t *qv = s_qv[wv], *qp = s_qp[wv], *qm = s_qm[wv];
    volatile uint32_t *pc = s_pc[wv], *pp = s_pp[wv];
    volatile uint64_t *pr = s_pr[wv];
    const int thr = cfg.thr;

    auto flush_runs = [&]() {   // whole wave
        const uint32_t n = *vn;
        uint32_t blo = 0, bhi = 0;
        if (lane == 0) {
            const unsigned long long b = atomicAdd(counters + 3, (unsigned long long)n);
            blo = (uint32_t)b; bhi = (uint32_t)(b >> 32);
        }
        blo = (uint32_t)__shfl((int)blo, 0, 64);
        bhi = (uint32_t)__shfl((int)bhi, 0, 64);
        const uint64_t base = ((uint64_t)bhi << 32) | blo;
        for (uint32_t k = lane; k < n; k += 64) {
            const uint64_t w = base + k;
            if (w < out.cap) { out.pidx[w] = vpi[k]; out.val[w] = vval[k]; }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) *vn = 0;
        __builtin_amdgcn_wave_barrier();
    };
    unsigned long long nhit = 0;
    // A queued leaf GROUP is a five-letter prefix (its value with the last digit 0) and the mask of the last letters that keep the k-mer similar.  64 groups at a
    // time (whole wave): the group's 20 presence bits are one or two words of the bitmap; only the k-mers that occur in the index are looked up in the offset
    // table (neighbouring entries), and the non-empty index ranges are staged.  (r04 probed the bitmap and the table once per similar k-mer, 3.2e10 times per
    // configs[3] @ 500 call, the last level one DFS step per letter.)
    auto drain_leaves = [&](uint32_t qn) {
        for (uint32_t b = 0; b < qn; b += 64) {
            const uint32_t i = b + lane;
            uint32_t m = 0, v0 = 0, pi = 0;
            if (i < qn) {
                v0 = qv[i]; pi = qp[i]; m = qm[i];
                if (kbits) {
                    const uint32_t w = v0 >> 5, sh = v0 & 31;
                    const uint64_t two = (uint64_t)kbits[w] | ((uint64_t)(sh > 32 - KA ? kbits[w + 1] : 0u) << 32);
                    m &= (uint32_t)(two >> sh);
                }
            }
            // FOUR k-mers of a group per round trip: the kernel's waves spend two thirds of their cycles waiting (PMC, profiles/r05/sim_runs_filter_pmc.txt),
            // mostly for these dependent offset-table loads - eight of them in flight per lane instead of two
            while (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
                uint32_t e0[4], n[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    e0[u] = 0; n[u] = 0;
                    if (m) {
                        const uint32_t v = v0 + (uint32_t)__builtin_ctz(m);
                        m &= m - 1;
                        e0[u] = koff[v];
                        n[u] = koff[v + 1];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t nn = n[u] - e0[u];           // (both 0 for a lane without a k-mer in this slot)
                    nhit += nn;
                    const uint64_t bm = __builtin_amdgcn_ballot_w64(nn != 0);
                    if (bm) {
                        if (*vn + 64 > RUN_STAGE) flush_runs();
                        const uint32_t slot = *vn + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
                        if (nn) { vval[slot] = ((uint64_t)nn << 32) | e0[u]; vpi[slot] = pi; }
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) *vn = *vn + (uint32_t)__popcll(bm);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
    };

    unsigned long long nsim = 0;
    const uint64_t npos = dk ? ((uint64_t)p1 - p0 + item_stride - 1) / item_stride : (uint64_t)p1 - p0;   // distinct mode: p1 - p0 items, sampled
    // one contiguous region per wave: the more positions a lane works through, the less the heaviest single position weighs
    const uint64_t nwaves = (uint64_t)gridDim.x * 4, region = (npos + nwaves - 1) / nwaves;
    {
        const uint64_t rg = (uint64_t)blockIdx.x * 4 + wv;
        // distinct mode: every wave walks the logical range [0, region) of its own stride-nwaves item sequence
        const uint64_t rbeg = dk ? 0 : min(rg * region, npos), rend = dk ? region : min(rbeg + region, npos);
        uint64_t next = rbeg;                      // wave-uniform cursor into the region
        uint32_t qn = 0;                           // wave-uniform fill of the leaf queue
        uint32_t head = 0, tail = 0;               // wave-uniform: pool of decoded positions (ring of POSQ entries)
        // per-lane DFS state
        bool idle = true, out_of_work = false;
        uint32_t cpack = 0, kpack = 0, pidx = 0, vcur = 0;
        uint32_t lc = 0;                           // leaves of the work item in hand
        bool has_item = false;
        int L = 0, scur = 0;
        uint64_t restpack = 0;                     // rest[m] (m = 1..6) in 8-bit fields, biased by 64
        for (;;) {
            // ---- decode the next 64 positions together (sequence search + letter loads: ~10 us of dependent loads that
            //      must not sit in front of every single hand-out) ----
            if (tail - head <= POSQ - 64 && next < rend) {
                const uint64_t mine = next + lane;
                // distinct mode: k-mer values come sorted and neighbours cost alike, so a wave takes every nwaves-th item
                // instead of a contiguous region
                const uint64_t item = dk ? (mine * nwaves + rg) * item_stride : mine;
                bool ok = false;
                uint32_t cp = 0;
                uint64_t rp = 0;
                if (mine < rend && item < (dk ? (uint64_t)p1 - p0 : npos)) {
                    uint32_t q, i, c[K];
                    bool valid;
                    if (dk) {
                        uint32_t v = dk[item];
#pragma unroll
                        for (int m = 0; m < K; m++) { c[m] = v % (uint32_t)KA; v /= (uint32_t)KA; }
                        valid = true;
                    } else {
                        valid = query_kmer_at(db, cfg, qbegin, qend, p0 + (uint32_t)mine, &q, &i, c);
                    }
                    if (valid) {
                        int rest = 0;
#pragma unroll
                        for (int L_ = K - 1; L_ >= 0; L_--) {                         // level L_ decides position K - 1 - L_
                            rp |= (uint64_t)(uint32_t)(rest + 64) << (8 * L_);      // field L_ holds the best score the levels behind L_ can still add
                            rest += tab.rowmax[c[K - 1 - L_]];
                            cp |= c[K - 1 - L_] << (5 * L_);
                        }
                        ok = rest >= thr;
                    }
                }
                const uint64_t om = __builtin_amdgcn_ballot_w64(ok);
                if (ok) {
                    const uint32_t slot = (tail + (uint32_t)__popcll(om & ((1ull << lane) - 1ull))) % POSQ;
                    pc[slot] = cp; pr[slot] = rp; pp[slot] = (uint32_t)item;
                }
                tail += (uint32_t)__popcll(om);
                next = min(next + 64, rend);
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            // ---- hand out decoded positions ----
            const uint64_t want = __builtin_amdgcn_ballot_w64(idle && !out_of_work);
            if (want) {
                const uint32_t take = head + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                if (idle && !out_of_work) {
                    if (take < tail) {
                        if (nsim_k && has_item) nsim_k[pidx] = lc;
                        lc = 0; has_item = true;
                        const uint32_t slot = take % POSQ;
                        cpack = pc[slot]; restpack = pr[slot]; pidx = pp[slot];
                        kpack = 0; L = 0; scur = 0; vcur = 0; idle = false;
                    } else if (next >= rend) out_of_work = true;       // pool empty and nothing left to decode
                }
                head = min(head + (uint32_t)__popcll(want), tail);
            }
            if (__builtin_amdgcn_ballot_w64(!idle) == 0) {
                if (next >= rend && head == tail) break;                       // region exhausted
                continue;
            }
            // ---- one DFS node per lane, branch-free: probe (L, k[L]); descend, emit a leaf, or step back to level L-1 ----
            // (three divergent paths would run one after the other in nearly every step; selects cost fewer issue slots)
            bool leaf = false;
            uint32_t leafv = 0, leafm = 0;
            {
                const int Lm = L > 0 ? L - 1 : 0;
                const uint32_t a = (cpack >> (5 * L)) & 31u, k = (kpack >> (5 * L)) & 31u;
                const uint32_t a2 = (cpack >> (5 * Lm)) & 31u, k2 = (kpack >> (5 * Lm)) & 31u;
                const uint32_t kk = k < (uint32_t)KA ? k : 0u, kk2 = k2 < (uint32_t)KA ? k2 : 0u;
                const int sc1 = tab.sc[a][kk], sc2 = tab.sc[a2][kk2];
                const uint32_t o1 = tab.ord[a][kk], o2 = tab.ord[a2][kk2];
                const uint32_t mulL = s_mul[L], mulM = s_mul[Lm];
                const int restn = (int)((restpack >> (8 * L)) & 0xffu) - 64;
                const int cand = scur + sc1;
                const bool act = !idle;
                const bool ok = act && k < (uint32_t)KA && cand + restn >= thr;
                const bool push = ok && L < K - 2, pop = act && !ok && L > 0;
                leaf = ok && L == K - 2;                                        // the last level is taken in one step: every letter that keeps cand + S >= thr
                leafv = vcur + o1 * mulL;
                {
                    const int need = min(max(thr - cand, -SIM_NEED_MAX), SIM_NEED_MAX);
                    const uint32_t al = (cpack >> (5 * (K - 1))) & 31u;
                    leafm = leaf ? tab.pmask[al][tab.lcnt[al][need + SIM_NEED_MAX]] : 0u;
                }
                lc += (uint32_t)__popc(leafm);
                idle = idle || (act && !ok && L == 0);
                scur = push ? cand : (pop ? scur - sc2 : scur);
                vcur = push ? leafv : (pop ? vcur - o2 * mulM : vcur);
                const uint32_t kp_leaf = kpack + (1u << (5 * (K - 2)));
                const uint32_t kp_push = kpack & ~(31u << (5 * (L + 1)));       // L + 1 <= K - 2 when push
                const uint32_t kp_pop = kpack + (1u << (5 * Lm));
                kpack = leaf ? kp_leaf : (push ? kp_push : (pop ? kp_pop : kpack));
                L = push ? L + 1 : (pop ? L - 1 : L);
            }
            const uint64_t lm = __builtin_amdgcn_ballot_w64(leaf);
            if (lm) {
                if (leaf) { const uint32_t slot = qn + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull)); qv[slot] = leafv; qp[slot] = pidx; qm[slot] = leafm; }
                qn += (uint32_t)__popcll(lm);
                nsim += (unsigned long long)__popc(leafm);
                if (qn > LEAFQ - 64) { __builtin_amdgcn_wave_barrier(); drain_leaves(qn); qn = 0; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        drain_leaves(qn);
        if (nsim_k && has_item) nsim_k[pidx] = lc;
    }
    __builtin_amdgcn_wave_barrier();
    flush_runs();
    for (int o = 32; o > 0; o >>= 1) { nsim += __shfl_down(nsim, o, 64); nhit += __shfl_down(nhit, o, 64); }
    if (lane == 0) {
        if (nsim) atomicAdd(counters + 0, nsim);
        if (nhit) atomicAdd(counters + 4, nhit);
    }
}

// ---- E2, distinct mode: the similar k-mers of a k-mer do not depend on where it occurs, and a batch of queries holds every
// k-mer several times (C2: 46.0 M positions, 14.9 M distinct k-mers; the ratio grows with the database).  So the DFS and the
// offset-table lookups run once per DISTINCT k-mer of the batch, the resulting run lists are sorted by k-mer rank (a third of
// the runs, 24-bit keys) and a copy kernel lays them out per query position, already in position order - the big sort of
// all runs by position is gone.
__global__ void __launch_bounds__(256) query_kmer_kernel(const DeviceDb db, KmerCfg cfg, uint32_t qbegin, uint32_t qend, uint32_t p0, uint32_t p1,
                                                         uint32_t *qk, uint8_t *flag) {
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < p1 - p0; idx += (uint64_t)gridDim.x * 256) {
        uint32_t q, i, c[K];
        uint32_t key = KMER_INVALID;
        if (query_kmer_at(db, cfg, qbegin, qend, p0 + (uint32_t)idx, &q, &i, c)) {
            uint32_t v = 0, mul = 1;
#pragma unroll
            for (int m = 0; m < K; m++) { v += c[m] * mul; mul *= KA; }
            key = v;
            flag[v] = 1;
        }
        qk[idx] = key;
    }
}

struct FlagToU32 {
    __host__ __device__ uint32_t operator()(uint8_t x) const { return (uint32_t)x; }
};

__global__ void __launch_bounds__(256) distinct_kmer_kernel(const uint8_t *flag, const uint32_t *kid, uint32_t *dk) {
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < KSPACE; v += (uint64_t)gridDim.x * 256)
        if (flag[v]) dk[kid[v]] = (uint32_t)v;
}

// first run of every k-mer rank in the runs sorted by rank (nd + 1 entries), and the rank's hit total
__global__ void __launch_bounds__(256) rank_offsets_kernel(const uint32_t *rk, uint32_t n_runs, uint32_t nd, uint32_t *roff) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k <= nd; k += (uint64_t)gridDim.x * 256) {
        uint32_t lo = 0, hi = n_runs;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (rk[mid] < (uint32_t)k) lo = mid + 1; else hi = mid;
        }
        roff[k] = lo;
    }
}

struct RankRec { uint32_t roff, nr, hits, nsim; };   // one 16-byte record per distinct k-mer: first run, runs, k-mer hits, similar k-mers

__global__ void __launch_bounds__(256) rank_rec_kernel(const uint32_t *roff, const uint64_t *val, const uint32_t *nsim_k, uint32_t nd, RankRec *rec) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < nd; k += (uint64_t)gridDim.x * 256) {
        uint64_t h = 0;
        const uint32_t r0 = roff[k], r1 = roff[k + 1];
        for (uint32_t r = r0; r < r1; r++) h += val[r] >> 32;
        RankRec x;
        x.roff = r0; x.nr = r1 - r0; x.nsim = nsim_k[k];
        x.hits = (uint32_t)min(h, (uint64_t)0xffffffffu);      // <= number of index entries < 2^32
        rec[k] = x;
    }
}

// per query position: runs, first run of its k-mer's list, k-mer hits; similar k-mers summed into counters[0]
__global__ void __launch_bounds__(256) position_runs_kernel(const uint32_t *qk, uint32_t npos, const uint32_t *kid, const RankRec *rec, uint32_t *nr,
                                                            uint32_t *src, uint32_t *ph, unsigned long long *counters) {
    unsigned long long sims = 0;
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < npos; idx += (uint64_t)gridDim.x * 256) {
        const uint32_t v = qk[idx];
        RankRec x = {0, 0, 0, 0};
        if (v != KMER_INVALID) x = rec[kid[v]];
        nr[idx] = x.nr; src[idx] = x.roff; ph[idx] = x.hits;
        sims += x.nsim;
    }
    for (int o = 32; o > 0; o >>= 1) sims += __shfl_down(sims, o, 64);
    if ((threadIdx.x & 63) == 0 && sims) atomicAdd(counters + 0, sims);
}

// per query: k-mer hits and runs (differences of the running sums over positions at the sequence boundaries)
__global__ void __launch_bounds__(256) query_totals_kernel(const uint32_t *off, uint32_t qbegin, uint32_t nq, uint32_t p0, const uint64_t *cumh,
                                                           const uint64_t *cumr, uint64_t *qh, uint64_t *qr) {
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (uint64_t)gridDim.x * 256) {
        const uint32_t a = off[qbegin + q] - p0, b = off[qbegin + q + 1] - p0;
        qh[q] = cumh[b] - cumh[a];
        qr[q] = cumr[b] - cumr[a];
    }
}

// a wave lays out the run lists of 64 consecutive positions of the batch [b0, b1): position order, runs of one position contiguous
__global__ void __launch_bounds__(256) position_expand_kernel(uint32_t b0, uint32_t b1, const uint32_t *nr, const uint32_t *src, const uint64_t *cumr,
                                                              const uint64_t *val, uint32_t *opidx, uint64_t *oval) {
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves = (uint64_t)gridDim.x * 4, w0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t rbase = cumr[b0];
    for (uint64_t base = b0 + w0 * 64; base < b1; base += nwaves * 64) {
        const uint64_t idx = base + lane;
        uint32_t n = 0, s = 0, dst = 0;
        if (idx < b1) {
            n = nr[idx];
            if (n) { s = src[idx]; dst = (uint32_t)(cumr[idx] - rbase); }
        }
        uint64_t m = __builtin_amdgcn_ballot_w64(n != 0);
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const uint32_t nj = (uint32_t)__shfl((int)n, j, 64), sj = (uint32_t)__shfl((int)s, j, 64), dj = (uint32_t)__shfl((int)dst, j, 64);
            for (uint32_t i = lane; i < nj; i += 64) {
                oval[dj + i] = val[sj + i];
                opidx[dj + i] = (uint32_t)(base + j - b0);
            }
        }
    }
}

// hit key: [ query - qbegin | target : tbits | diagonal + dbias : dbits ]; the field widths follow the database
// (number of sequences, longest sequence) so that the radix sort touches as few digits as possible
struct KeyFmt {
    int dbits, tbits;
    int dbias;
    // compact mode (the common case): when (target - tbase) and the diagonal fit 32 bits together, index entries are
    // u32 [target - tbase | position : dbits - 1] and the per-query (target, diagonal) keys of the double-hit filter are
    // u32 [target - tbase | diagonal + dbias : dbits] - half the bytes of the wide u64 forms on every gather and stream
    int compact;
    uint32_t tbase;
};

// one index entry -> (global target id, position)
template <bool C>
__device__ __forceinline__ void ent_decode(const void *ent, uint32_t idx, const KeyFmt &fmt, uint32_t &t, uint32_t &pos) {
    if (C) {
        const uint32_t e = ((const uint32_t *)ent)[idx];
        t = (e >> (fmt.dbits - 1)) + fmt.tbase;
        pos = e & ((1u << (fmt.dbits - 1)) - 1u);
    } else {
        const uint64_t e = ((const uint64_t *)ent)[idx];
        t = (uint32_t)(e >> 16);
        pos = (uint32_t)(e & 0xFFFF);
    }
}

// ---- E2 pass 2: expand runs into hit keys, load-balanced ----
// A workgroup takes 256 runs, scans their lengths, reserves the tile's key range with one atomic and then lets
// thread k write key k of the tile (binary search in the LDS prefix): loads of index entries touch a handful of
// lines per wave and the key stores are fully coalesced.
__global__ void __launch_bounds__(256) expand_kernel(const DeviceDb db, uint32_t qbegin, uint32_t qend, uint32_t p0, RunList runs,
                                                     uint64_t n_runs, const void *ent, KeyFmt fmt,
                                                     unsigned long long *key_cursor, uint64_t *keys, uint64_t key_cap) {
    __shared__ uint64_t s_pref[257];
    __shared__ uint64_t s_qb[256];
    __shared__ uint32_t s_e0[256];
    __shared__ int32_t s_i[256];
    __shared__ uint64_t s_wsum[4];
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (uint64_t tile = blockIdx.x; tile * 256 < n_runs; tile += gridDim.x) {
        const uint64_t r = tile * 256 + tid;
        uint64_t c = 0;
        if (r < n_runs) {
            const uint64_t rv = runs.val[r];
            c = rv >> 32;
            const uint32_t p = p0 + runs.pidx[r];
            const uint32_t s = find_seq(db.off, qbegin, qend, p);
            s_e0[tid] = (uint32_t)rv;
            s_i[tid] = (int32_t)(p - db.off[s]) + fmt.dbias;
            s_qb[tid] = (uint64_t)(s - qbegin) << (fmt.tbits + fmt.dbits);
        }
        uint64_t inc = c;   // inclusive scan inside the wave, then across the 4 waves
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_wsum[wv] = inc;
        __syncthreads();
        uint64_t woff = 0;
        for (int w = 0; w < wv; w++) woff += s_wsum[w];
        s_pref[tid] = woff + inc - c;
        if (tid == 255) {
            const uint64_t total = woff + inc;
            s_pref[256] = total;
            s_base = total ? atomicAdd(key_cursor, (unsigned long long)total) : 0ull;
        }
        __syncthreads();
        const uint64_t total = s_pref[256], base = s_base;
        for (uint64_t k = tid; k < total; k += 256) {
            int lo = 0, hi = 256;          // last run with pref <= k
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_pref[mid] <= k) lo = mid; else hi = mid;
            }
            uint32_t et, epos;
            if (fmt.compact) ent_decode<true>(ent, s_e0[lo] + (uint32_t)(k - s_pref[lo]), fmt, et, epos);
            else ent_decode<false>(ent, s_e0[lo] + (uint32_t)(k - s_pref[lo]), fmt, et, epos);
            const uint64_t key = s_qb[lo] | ((uint64_t)et << fmt.dbits) | (uint64_t)(s_i[lo] - (int32_t)epos);
            if (base + k < key_cap) keys[base + k] = key;
        }
        __syncthreads();
    }
}

// ---- E2 pass 2 (min_diag_hits >= 2): expand + double-hit filter, one workgroup per query ----
// Only hits that share (target, diagonal) with another hit of the same query can make a candidate, and they are a
// few percent of all hits.  With the runs sorted by query position a workgroup owns one query:
//   sweep 1 expands the query's runs (load-balanced: thread k takes KPT consecutive keys of the tile by one binary
//           search in the LDS prefix of the run lengths), records every (target, diagonal) hash in two LDS bitmaps
//           (seen once / seen twice, two hash positions per key) and STREAMS the (target, diagonal) keys it computed
//           into the query's region of the key buffer (coalesced 16-byte stores);
//   sweep 2 reads that stream back — a coalesced scan instead of a second round of binary searches and index gathers
//           (the gathers were the kernel's critical path: 69 % of its cycles waited on them, profiles/r2b) — keeps the
//           keys whose two hash positions were both seen twice, and compacts them in place;
//   level 2 repeats the once/twice test over the survivors alone with independent hash functions (below).
// Every key of a real multi-hit diagonal survives, plus a few collisions which the exact diagonal count after the sort
// discards again.  The region is sized by the query's exact hit total (from sim_runs), so nothing can overflow.
constexpr int FB_LOG2 = 19;            // bits per bitmap: 2 x 64 KiB of the CU's 160 KiB LDS
constexpr int FT = 1024;               // threads per workgroup
constexpr int KPT = 4;                 // consecutive keys per thread and binary search
// runs per tile = RPT x FT.  RPT = 2 (r4): half as many tile hand-overs (three workgroup barriers and the drain of 16 waves each) per
// query; the prefix search takes 11 probes instead of 10
constexpr int FILTER_RPT = 2;
constexpr size_t filter_lds(int rpt) { return 2 * ((size_t)1 << FB_LOG2) / 8 + ((size_t)rpt * FT + 1) * 4 + (size_t)rpt * FT * 8 + 16 * 4 + 64; }

// first run of every query of the batch (runs are sorted by query position): qr[i] = lower_bound(rpidx, off[qbegin + i] - p0),
// i = 0..nq.  One thread per query here instead of two serial ~25-step searches at the head of every filter workgroup
// (those cost ~20 us of the ~100 us a query spends in the filter kernel).
__global__ void __launch_bounds__(256) run_range_kernel(const DeviceDb db, uint32_t qbegin, uint32_t nq, uint32_t p0, const uint32_t *rpidx,
                                                        uint64_t n_runs, uint64_t *qr) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i <= nq; i += gridDim.x * 256) {
        const uint32_t want = db.off[qbegin + i] - p0;
        uint64_t lo = 0, hi = n_runs;
        while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (rpidx[m] < want) lo = m + 1; else hi = m; }
        qr[i] = lo;
    }
}

// launch order of the filter workgroups: queries with the most runs first (longest-processing-time-first: a batch then ends
// with many short queries instead of a few long ones; the tail was ~20 % of the kernel)
__global__ void __launch_bounds__(256) run_order_key_kernel(uint32_t nq, const uint64_t *qr, uint32_t *key, uint32_t *idx) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) {
        const uint64_t c = qr[i + 1] - qr[i];
        key[i] = 0xFFFFFFFFu - (uint32_t)min<uint64_t>(c, 0xFFFFFFFFull);
        idx[i] = i;
    }
}

template <bool C> struct TdType { using type = uint64_t; };
template <> struct TdType<true> { using type = uint32_t; };

// (Measured and removed, profiles/r03_filter_ab.log: 1024-run tiles — 2.5 % slower; a blocked Bloom filter with both hash positions of a key in one
// 64-bit LDS word, one ds_or_rtn_b64 instead of two dependent 32-bit atomic pairs — no gain: the kernel is not bound by its LDS atomics.)
template <bool C>
__global__ void __launch_bounds__(FT) filter_kernel(const DeviceDb db, uint32_t qbegin, uint32_t p0, const uint32_t *rpidx,
                                                    const uint64_t *rval, const uint64_t *qr, const uint32_t *order, const void *ent, KeyFmt fmt,
                                                    unsigned long long *region_cursor, void *region_v, uint64_t region_cap,
                                                    uint64_t *qbase, uint32_t *qsurv) {
    using TD = typename TdType<C>::type;
    struct __attribute__((aligned(4))) TD4 { TD v[KPT]; };        // KPT keys of one thread: dword-aligned vector access
    TD *region = (TD *)region_v;
    extern __shared__ __attribute__((aligned(16))) uint32_t f_lds[];
    constexpr int BW = 1 << (FB_LOG2 - 5);          // words per bitmap
    constexpr int RPT = FILTER_RPT;
    constexpr int TR = RPT * FT;                    // runs per tile
    uint32_t *B1 = f_lds, *B2 = f_lds + BW;
    uint2 *s_run = (uint2 *)(f_lds + 2 * BW);                   // TR: {first index entry - prefix, query position + bias}: ONE 8-byte read per key
    uint32_t *s_pref = (uint32_t *)(s_run + TR);                // TR + 1 (hits of one query < 2^32: checked by the host)
    uint32_t *s_wsum = s_pref + TR + 1;                         // 16
    uint64_t *s_misc = (uint64_t *)(((uintptr_t)(s_wsum + 16) + 7) & ~(uintptr_t)7);   // [0] r0, [1] r1, [2] region base, [3] cursors (2 x u32)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t qi = order[blockIdx.x];                      // query of this workgroup (index inside the batch)
    const uint32_t q = qbegin + qi;
    const uint32_t plo = db.off[q] - p0;
    const uint64_t r0 = qr[qi], r1 = qr[qi + 1];
    TD *surv = region + region_cap;                             // second area of the same size: survivors of sweep 2
    for (int w = tid; w < 2 * BW; w += FT) f_lds[w] = 0;
    if (r0 == r1) {
        if (tid == 0) { qbase[qi] = 0; qsurv[qi] = 0; }
        return;
    }

    // loads one tile of runs, leaves the exclusive prefix of their lengths in s_pref[0..FT] (s_pref[FT] = total)
    // the run of the NEXT tile is fetched while the current one is expanded (the loads' latency would otherwise sit between
    // two barriers with nothing else to run: one workgroup per CU)
    uint64_t pf_rv[RPT] = {};
    uint32_t pf_pi[RPT] = {};
    auto prefetch_tile = [&](uint64_t tile) {      // thread tid owns runs tile + RPT * tid .. + RPT - 1 (consecutive: the prefix stays a plain scan)
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const uint64_t r = tile + (uint64_t)RPT * tid + j;
            if (r < r1) { pf_rv[j] = rval[r]; pf_pi[j] = rpidx[r]; }
        }
    };
    auto load_tile = [&](uint64_t tile) {
        uint32_t c[RPT], csum = 0;
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const uint64_t r = tile + (uint64_t)RPT * tid + j;
            c[j] = r < r1 ? (uint32_t)(pf_rv[j] >> 32) : 0u;
            csum += c[j];
        }
        uint32_t inc = csum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_wsum[wv] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wv; w++) woff += s_wsum[w];
        uint32_t pre = woff + inc - csum;
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            s_pref[RPT * tid + j] = pre;
            s_run[RPT * tid + j] = make_uint2((uint32_t)pf_rv[j] - pre, (uint32_t)((int32_t)(pf_pi[j] - plo) + fmt.dbias));
            pre += c[j];
        }
        if (tid == FT - 1) s_pref[TR] = pre;
        __syncthreads();
    };
    // keys k4 .. k4+KPT-1 of the current tile: one binary search finds the run of the first key; every run holds at least
    // one key, so each following key is in the same run or the next one (a branch-free step).  locate() is pure LDS work
    // with a fixed trip count, so the two groups a thread handles per iteration interleave and their 2 x KPT index loads
    // are in flight together (the kernel is bound by the latency of search -> gather -> mark, one workgroup per CU).
    auto locate = [&](uint32_t k4, uint32_t T, uint32_t (&idx)[KPT], int32_t (&si)[KPT]) -> int {
        int lo = 0;
#pragma unroll
        for (int step = TR / 2; step >= 1; step >>= 1)            // TR = 2^10 / 2^11: ten / eleven probes
            if (s_pref[lo + step] <= k4) lo += step;
        const int n = k4 >= T ? 0 : (T - k4 < (uint32_t)KPT ? (int)(T - k4) : KPT);
#pragma unroll
        for (int i = 0; i < KPT; i++) {
            if (i > 0 && i < n) lo += (s_pref[lo + 1] <= k4 + i) ? 1 : 0;
            const uint2 rr = s_run[lo];
            idx[i] = rr.x + (k4 + i);
            si[i] = (int32_t)rr.y;
        }
#pragma unroll
        for (int i = 1; i < KPT; i++) if (i >= n) { idx[i] = idx[0]; si[i] = si[0]; }
        return n;
    };
    auto gather = [&](const uint32_t (&idx)[KPT], typename std::conditional<C, uint32_t, uint64_t>::type (&e)[KPT]) {
#pragma unroll
        for (int i = 0; i < KPT; i++) {
            if (C) e[i] = ((const uint32_t *)ent)[idx[i]];
            else e[i] = ((const uint64_t *)ent)[idx[i]];
        }
    };
    auto to_td = [&](const typename std::conditional<C, uint32_t, uint64_t>::type (&e)[KPT], const int32_t (&si)[KPT], TD (&td)[KPT]) {
        const uint32_t pm = (1u << (fmt.dbits - 1)) - 1u;
#pragma unroll
        for (int i = 0; i < KPT; i++) {
            if (C) td[i] = (TD)((((uint32_t)e[i] >> (fmt.dbits - 1)) << fmt.dbits) | (uint32_t)(si[i] - (int32_t)((uint32_t)e[i] & pm)));
            else td[i] = (TD)((((uint64_t)e[i] >> 16) << fmt.dbits) | (uint64_t)(si[i] - (int32_t)((uint64_t)e[i] & 0xFFFF)));
        }
    };
    // two independent hash positions per (target, diagonal): a key survives only if BOTH were seen twice (a Bloom
    // filter with k = 2: collisions let ~2 % of the single hits through instead of ~11 %, which is what the sort pays for)
    auto slot_of = [&](uint64_t td, uint32_t &h2) -> uint32_t {
        const uint32_t x = (uint32_t)td * 0x9E3779B1u ^ (uint32_t)(td >> 32) * 0x85EBCA6Bu;
        h2 = ((x ^ (x >> 15)) * 0x846CA68Bu) >> (32 - FB_LOG2);
        return (x * 0x2C1B3C6Du) >> (32 - FB_LOG2);
    };
    auto mark1 = [&](uint64_t td) {
        uint32_t g;
        const uint32_t h = slot_of(td, g), bit = 1u << (h & 31), gbit = 1u << (g & 31);
        const uint32_t old = atomicOr(&B1[h >> 5], bit);
        if (old & bit) atomicOr(&B2[h >> 5], bit);
        const uint32_t gold = atomicOr(&B1[g >> 5], gbit);
        if (gold & gbit) atomicOr(&B2[g >> 5], gbit);
    };
    auto twice1 = [&](uint64_t td) -> uint32_t {
        uint32_t g;
        const uint32_t h = slot_of(td, g);
        return ((B2[h >> 5] >> (h & 31)) & (B2[g >> 5] >> (g & 31))) & 1u;
    };

    // the query's region: its exact hit total is the sum of its run lengths; reserve it first (rounded up to KPT keys)
    {
        uint64_t part = 0;
        for (uint64_t r = r0 + tid; r < r1; r += FT) part += rval[r] >> 32;
        for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
        uint64_t *s_part = (uint64_t *)s_pref;      // not in use yet
        if (lane == 0) s_part[wv] = part;
        __syncthreads();
        if (tid == 0) {
            uint64_t tot = 0;
            for (int w = 0; w < FT / 64; w++) tot += s_part[w];
            s_misc[2] = atomicAdd(region_cursor, (unsigned long long)((tot + KPT - 1) / KPT * KPT));
            ((uint32_t *)&s_misc[3])[0] = 0;
            ((uint32_t *)&s_misc[3])[1] = 0;
        }
        __syncthreads();
    }
    const uint64_t base = s_misc[2];
    uint32_t *cur = (uint32_t *)&s_misc[3];
    uint32_t total = 0;
    prefetch_tile(r0);
    for (uint64_t tile = r0; tile < r1; tile += TR) {
        load_tile(tile);
        prefetch_tile(tile + TR);
        const uint32_t T = s_pref[TR];
        using ET = typename std::conditional<C, uint32_t, uint64_t>::type;
        auto mark_store = [&](TD4 &td, int n, uint32_t k4) {
#pragma unroll
            for (int i = 0; i < KPT; i++) {
                if (i >= n) { td.v[i] = td.v[0]; continue; }
                mark1(td.v[i]);
            }
            const uint64_t w = base + total + k4;
            if (n && w + KPT <= region_cap) {
                if (n == KPT) *(TD4 *)(region + w) = td;
                else {
#pragma unroll
                    for (int i = 0; i < KPT; i++) if (i < n) region[w + i] = td.v[i];     // (a runtime trip count would put td into scratch)
                }
            }
        };
        for (uint32_t k4 = (uint32_t)KPT * tid; k4 < T; k4 += 2u * KPT * FT) {
            const uint32_t k4b = k4 + (uint32_t)KPT * FT;
            uint32_t ia[KPT], ib[KPT];
            int32_t sa[KPT], sb[KPT];
            const int na = locate(k4, T, ia, sa);
            const int nb = locate(k4b < T ? k4b : k4, T, ib, sb);
            ET ea[KPT], eb[KPT];
            gather(ia, ea);
            gather(ib, eb);
            TD4 ta, tb;
            to_td(ea, sa, ta.v);
            mark_store(ta, na, k4);
            to_td(eb, sb, tb.v);
            mark_store(tb, k4b < T ? nb : 0, k4b);
        }
        total += T;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // sweep 2: stream the keys back, keep those whose two positions were both seen twice and append them to the query's
    // slice of the survivor area (same offsets as its region): no barrier and no in-place hazard, so the loads of successive
    // iterations overlap.
    const uint32_t n0 = (uint32_t)min<uint64_t>(total, region_cap > base ? region_cap - base : 0);
    const uint32_t n0r = n0;                 // (a wave's lanes enter the loop together: their k4 differ by < 64 * KPT <= the loop stride; the ballots below only need the lanes that are in)
    for (uint32_t k4 = (uint32_t)KPT * tid; k4 < n0r; k4 += 2u * KPT * FT) {
        // two groups of KPT keys per thread and iteration: both loads are in flight before either is tested
        TD4 td[2] = {};
        int nn[2];
#pragma unroll
        for (int g2 = 0; g2 < 2; g2++) {
            const uint32_t k = k4 + (uint32_t)g2 * KPT * FT;
            nn[g2] = k >= n0 ? 0 : (n0 - k < (uint32_t)KPT ? (int)(n0 - k) : KPT);
            if (nn[g2] == KPT) td[g2] = *(const TD4 *)(region + base + k);
            else {
#pragma unroll
                for (int i = 0; i < KPT; i++) if (i < nn[g2]) td[g2].v[i] = region[base + k + i];
            }
        }
        uint32_t keep = 0;                      // bit 4 * g2 + i: key i of group g2 survives
#pragma unroll
        for (int g2 = 0; g2 < 2; g2++)
#pragma unroll
            for (int i = 0; i < KPT; i++)
                if (i < nn[g2]) keep |= twice1(td[g2].v[i]) << (KPT * g2 + i);
        // wave-level compaction by ballots (the order of the survivors is irrelevant: they get sorted): rank of key j of
        // this lane = survivors of keys < j over the whole wave + survivors of key j in the lanes below
        uint32_t rank[2 * KPT];
        uint32_t wave_total = 0;
#pragma unroll
        for (int j = 0; j < 2 * KPT; j++) {
            const uint64_t m = __builtin_amdgcn_ballot_w64((keep >> j) & 1u);
            rank[j] = wave_total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            wave_total += (uint32_t)__popcll(m);
        }
        if (wave_total) {
            uint32_t s0 = 0;
            if (lane == 0) s0 = atomicAdd(cur, wave_total);
            s0 = (uint32_t)__shfl((int)s0, 0, 64);
#pragma unroll
            for (int j = 0; j < 2 * KPT; j++)
                if ((keep >> j) & 1u) surv[base + s0 + rank[j]] = td[j / KPT].v[j % KPT];
        }
    }
    // ---- second level: the same once / twice test over the SURVIVORS only, with independent hash functions ----
    // Long queries load the bitmaps so heavily that most first-level survivors are single hits whose two positions were
    // set by other keys (C2: 0.55 G survivors for ~0.03 G keys of real double-hit diagonals).  Every key of a diagonal with
    // >= 2 hits survives level 1 together with its twins, so repeating the test over the ~10x shorter survivor list (a
    // coalesced read of the query's own slice, no index lookups) keeps all of them and drops nearly all the rest; the final
    // keys go back to the head of the query's region.
    __threadfence_block();
    __syncthreads();
    const uint32_t n1 = *cur;
    for (int w = tid; w < 2 * BW; w += FT) f_lds[w] = 0;
    uint32_t *cur2 = cur + 1;
    __syncthreads();
    auto slot2_of = [&](uint64_t key, uint32_t &h2) -> uint32_t {
        const uint32_t x = (uint32_t)key * 0xC2B2AE35u ^ (uint32_t)(key >> 32) * 0x27D4EB2Fu;
        h2 = ((x ^ (x >> 13)) * 0x165667B1u) >> (32 - FB_LOG2);
        return (x * 0x9E3779B1u) >> (32 - FB_LOG2);
    };
    for (uint32_t k = tid; k < n1; k += FT) {
        uint32_t g;
        const uint32_t h = slot2_of(surv[base + k], g), bit = 1u << (h & 31), gbit = 1u << (g & 31);
        const uint32_t old = atomicOr(&B1[h >> 5], bit);
        if (old & bit) atomicOr(&B2[h >> 5], bit);
        const uint32_t gold = atomicOr(&B1[g >> 5], gbit);
        if (gold & gbit) atomicOr(&B2[g >> 5], gbit);
    }
    __syncthreads();
    const uint32_t n1r = (n1 + 63) / 64 * 64;
    for (uint32_t k = tid; k < n1r; k += FT) {
        TD key = 0;
        bool keep = false;
        if (k < n1) {
            key = surv[base + k];
            uint32_t g;
            const uint32_t h = slot2_of(key, g);
            keep = ((B2[h >> 5] >> (h & 31)) & (B2[g >> 5] >> (g & 31))) & 1u;
        }
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        if (m) {
            uint32_t s0 = 0;
            if (lane == 0) s0 = atomicAdd(cur2, (uint32_t)__popcll(m));
            s0 = (uint32_t)__shfl((int)s0, 0, 64);
            if (keep) region[base + s0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
        }
    }
    __syncthreads();
    if (tid == 0) { qbase[qi] = base; qsurv[qi] = *cur2; }
}

// moves every query's surviving (target, diagonal) keys from its region to a dense array of full sort keys
// [ query - qbegin | target | diagonal + dbias ]
template <bool C>
__global__ void __launch_bounds__(256) compact_kernel(const void *region_v, const uint64_t *qbase, const uint32_t *qsurv,
                                                      const uint64_t *soff, uint32_t nq, KeyFmt fmt, uint64_t *out) {
    using TD = typename TdType<C>::type;
    const TD *region = (const TD *)region_v;
    const uint64_t dmask = (1ull << fmt.dbits) - 1;
    for (uint32_t b = blockIdx.x; b < nq; b += gridDim.x) {
        const uint64_t src = qbase[b], dst = soff[b];
        const uint32_t n = qsurv[b];
        const uint64_t qbits = (uint64_t)b << (fmt.tbits + fmt.dbits);
        for (uint32_t k = threadIdx.x; k < n; k += 256) {
            const uint64_t td = region[src + k];
            out[dst + k] = C ? (qbits | (((td >> fmt.dbits) + fmt.tbase) << fmt.dbits) | (td & dmask)) : (qbits | td);
        }
    }
}

// one pass over the sorted hit keys: per (query,target) group the diagonals are run-length-counted and the best one
// (count desc, diagonal asc) is kept.  Candidates are staged per wave in LDS and
// appended in blocks of >= 64 behind one global atomic (one atomic per wave ballot serialised on a single
// address: +90 ms per step, profiles/r1g); their order is irrelevant (E4 sorts on a unique key).
//
// Symmetric passes (mirror_q0 < UINT32_MAX; Engine::prefilter, all-vs-all over several target chunks under a symmetric matrix): the k-mer hit
// relation is symmetric — query position i hits target position j iff S(kmer_q(i), kmer_t(j)) >= thr — so the hits of (t, q) are those of
// (q, t) with the diagonal negated.  A pass over target chunk c then only matches the queries from chunk c onwards, and every group (q, t)
// whose query lies BEHIND the chunk (q >= mirror_q0) also emits the candidate of the pair the other way round, (t, q), under ITS tie-break:
// most hits, then the smallest diagonal of (t, q) = the LARGEST diagonal of (q, t).
// Long groups: a sequence against itself (and close homologs) gives groups of hundreds to thousands of keys with one run — diagonal 0 — as long as
// the sequence; a lane walking such a group or run key by key (dependent loads) was the kernel's whole duration (~2.5 ms per launch at configs[1]
// whatever the other 99 % of the keys cost).  A group that covers the whole NEXT window is therefore only flagged here (wflag[window] = lane + 1)
// and evaluated by diag_long_kernel, a wave per long group; every serial walk left in this kernel is bounded by 64 keys.
__global__ void __launch_bounds__(256) diag_select_kernel(const uint64_t *keys, uint64_t n, int min_hits, KeyFmt fmt, uint32_t qbegin,
                                                          unsigned long long *n_cand, uint64_t cap,
                                                          uint32_t *cq, uint32_t *ct, int32_t *cd, uint32_t mirror_q0, uint8_t *wflag) {
    // candidates are staged per wave and appended behind ONE global cursor in blocks of >= 64 (a 512-entry stage was measured SLOWER, 16 vs 11 ms:
    // the appends are not what bounds the kernel — the serial walks of the long groups were, see below)
    constexpr int DS_SLOTS = 128, DS_DRAIN = 64;
    __shared__ uint32_t s_q[4][DS_SLOTS], s_t[4][DS_SLOTS];
    __shared__ int32_t s_d[4][DS_SLOTS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t dmask = (1ull << fmt.dbits) - 1, tmask = (1ull << fmt.tbits) - 1;
    const uint64_t nround = (n + 255) / 256 * 256;   // whole waves stay in the loop (ballot below)
    uint32_t staged = 0;                             // wave-uniform
    auto drain = [&]() {
        uint32_t blo = 0, bhi = 0;
        if (lane == 0) {
            const unsigned long long b = atomicAdd(n_cand, (unsigned long long)staged);
            blo = (uint32_t)b; bhi = (uint32_t)(b >> 32);
        }
        blo = (uint32_t)__shfl((int)blo, 0, 64);
        bhi = (uint32_t)__shfl((int)bhi, 0, 64);
        const uint64_t base = ((uint64_t)bhi << 32) | blo;
        for (uint32_t k = lane; k < staged; k += 64) {
            const uint64_t w = base + k;
            if (w < cap) { cq[w] = s_q[wv][k]; ct[w] = s_t[wv][k]; cd[w] = s_d[wv][k]; }
        }
        staged = 0;
        __builtin_amdgcn_wave_barrier();
    };
    // A wave looks at 64 consecutive keys at a time (one coalesced load; r04 — until then the first key of every group walked its group alone with
    // dependent loads while the other lanes of the wave idled: 9.4 ms for 0.22 G keys at configs[1], ~0.2 TB/s).  Run starts (key differs from its
    // predecessor) and group starts come from two ballots; a run's length is the distance to the next run start; the group's first lane folds the
    // (length, diagonal) pairs of its runs out of LDS.  Only the ONE run and the ONE group that reach past the wave's 64 keys are finished serially.
    __shared__ uint64_t s_v1[4][64], s_v2[4][64];
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nround; i += (uint64_t)gridDim.x * 256) {
        bool cand = false, deferred = false;
        int best_d = 0, last_d = 0;
        uint64_t grp = 0;
        const uint64_t wbase = i - (uint64_t)lane;                    // first key of this wave's window
        const bool valid = i < n;
        const uint64_t k = valid ? keys[i] : ~0ull;
        uint64_t kp = (uint64_t)__shfl_up((unsigned long long)k, 1, 64);
        if (lane == 0) kp = (valid && i > 0) ? keys[i - 1] : ~k;
        const bool run_start = valid && kp != k;
        const bool grp_start = valid && (i == 0 || (kp >> fmt.dbits) != (k >> fmt.dbits));
        const uint64_t starts = __builtin_amdgcn_ballot_w64(run_start), gstarts = __builtin_amdgcn_ballot_w64(grp_start);
        const uint64_t above = lane == 63 ? 0ull : (~0ull << (lane + 1));
        const uint64_t wend = wbase + 64 < n ? wbase + 64 : n;        // first key behind the window (or the end of the list)
        // the window's last group — the only one that can reach behind it — is LONG if it covers the whole next window: decided first (one load by its
        // first lane), so that the window's last run, which belongs to that group, is not walked for nothing
        if (grp_start && !(gstarts & above) && wend < n) {
            const uint64_t far = wend + 63 < n ? wend + 63 : n - 1;
            deferred = (keys[far] >> fmt.dbits) == (k >> fmt.dbits);
        }
        const bool long_tail = __builtin_amdgcn_ballot_w64(deferred) != 0;
        uint64_t v1 = 0, v2 = 0;
        if (run_start) {
            const uint64_t nx = starts & above;
            uint64_t cnt;
            if (nx) cnt = (uint64_t)(__builtin_ctzll(nx) - lane);
            else {                                                     // the window's last run: it may go on behind the window, for < 64 keys
                uint64_t e = wend;                                     // (otherwise its group is long and nothing here is used)
                const uint64_t lim = wend + 64 < n ? wend + 64 : n;
                if (!long_tail) while (e < lim && keys[e] == k) e++;
                cnt = e - i;
            }
            const uint64_t d = k & dmask;                              // diagonal + bias
            v1 = (cnt << 32) | (0xFFFFFFFFull - d);                    // max: most hits, then the SMALLEST diagonal
            v2 = (cnt << 32) | d;                                      // max: most hits, then the LARGEST diagonal (the pair the other way round)
        }
        s_v1[wv][lane] = v1; s_v2[wv][lane] = v2;
        __builtin_amdgcn_wave_barrier();
        if (grp_start) {
            grp = k >> fmt.dbits;
            const uint64_t ng = gstarts & above;
            const int gend = ng ? __builtin_ctzll(ng) : 64;            // this group's lanes inside the window: [lane, gend)
            uint64_t b1 = 0, b2 = 0;
            uint64_t runs = starts & (~0ull << lane) & (gend == 64 ? ~0ull : ((1ull << gend) - 1ull));
            while (runs) {
                const int r = __builtin_ctzll(runs);
                runs &= runs - 1;
                const uint64_t a = s_v1[wv][r], c2 = s_v2[wv][r];
                b1 = a > b1 ? a : b1; b2 = c2 > b2 ? c2 : b2;
            }
            if (gend == 64 && wend < n && !deferred) {                 // the group ends inside the next window: the rest of it, run by run
                uint64_t e = wend;
                // skip what is left of the window's last run (already counted by its own lane), then walk the following runs of the group
                const uint64_t kw = keys[wend - 1];
                while (e < n && keys[e] == kw) e++;
                while (e < n && (keys[e] >> fmt.dbits) == grp) {
                    const uint64_t kb = keys[e];
                    uint64_t f = e + 1;
                    while (f < n && keys[f] == kb) f++;
                    const uint64_t cnt = f - e, d = kb & dmask;
                    const uint64_t a = (cnt << 32) | (0xFFFFFFFFull - d), c2 = (cnt << 32) | d;
                    b1 = a > b1 ? a : b1; b2 = c2 > b2 ? c2 : b2;
                    e = f;
                }
            }
            cand = !deferred && (int)(b1 >> 32) >= min_hits;
            best_d = (int)(0xFFFFFFFFull - (b1 & 0xFFFFFFFFull)) - fmt.dbias;
            last_d = (int)(b2 & 0xFFFFFFFFull) - fmt.dbias;
        }
        __builtin_amdgcn_wave_barrier();                               // the LDS rows are rewritten in the next round
        {   // at most one group of a window can be long (it reaches the window's end): one flag byte per window
            const uint64_t dm = __builtin_amdgcn_ballot_w64(deferred);
            if (lane == 0 && wbase < n) wflag[wbase >> 6] = dm ? (uint8_t)(__builtin_ctzll(dm) + 1) : (uint8_t)0;
        }
        const uint32_t gq = qbegin + (uint32_t)(grp >> fmt.tbits), gt = (uint32_t)(grp & tmask);
        const uint64_t m = __builtin_amdgcn_ballot_w64(cand);
        if (m) {
            if (cand) {
                const uint32_t k = staged + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                s_q[wv][k] = gq;
                s_t[wv][k] = gt;
                s_d[wv][k] = best_d;
            }
            staged += (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (staged >= DS_DRAIN) drain();
        }
        // the pair the other way round (a second round of at most 64 entries: a stage that was below DS_DRAIN cannot overflow)
        const bool mcand = cand && gq >= mirror_q0;
        const uint64_t mm = __builtin_amdgcn_ballot_w64(mcand);
        if (mm) {
            if (mcand) {
                const uint32_t k = staged + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                s_q[wv][k] = gt;
                s_t[wv][k] = gq;
                s_d[wv][k] = -last_d;
            }
            staged += (uint32_t)__popcll(mm);
            __builtin_amdgcn_wave_barrier();
            if (staged >= DS_DRAIN) drain();
        }
    }
    if (staged) drain();
}

// the long groups flagged by diag_select_kernel: a wave per group scans it 64 keys at a time; a run that reaches the end of a window is carried
// (key, length so far) into the next one, so nothing is walked key by key.  Same result rule, same staging of the candidates.
__global__ void __launch_bounds__(256) diag_long_kernel(const uint64_t *keys, uint64_t n, int min_hits, KeyFmt fmt, uint32_t qbegin,
                                                        unsigned long long *n_cand, uint64_t cap,
                                                        uint32_t *cq, uint32_t *ct, int32_t *cd, uint32_t mirror_q0, const uint8_t *wflag) {
    __shared__ uint32_t s_q[4][128], s_t[4][128];
    __shared__ int32_t s_d[4][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t dmask = (1ull << fmt.dbits) - 1, tmask = (1ull << fmt.tbits) - 1;
    const uint64_t nwin = (n + 63) >> 6;
    uint32_t staged = 0;
    auto drain = [&]() {
        uint32_t blo = 0, bhi = 0;
        if (lane == 0) {
            const unsigned long long b = atomicAdd(n_cand, (unsigned long long)staged);
            blo = (uint32_t)b; bhi = (uint32_t)(b >> 32);
        }
        blo = (uint32_t)__shfl((int)blo, 0, 64);
        bhi = (uint32_t)__shfl((int)bhi, 0, 64);
        const uint64_t base = ((uint64_t)bhi << 32) | blo;
        for (uint32_t k = lane; k < staged; k += 64) {
            const uint64_t w = base + k;
            if (w < cap) { cq[w] = s_q[wv][k]; ct[w] = s_t[wv][k]; cd[w] = s_d[wv][k]; }
        }
        staged = 0;
        __builtin_amdgcn_wave_barrier();
    };
    auto wmax = [&](uint64_t v) -> uint64_t {
        for (int o = 32; o > 0; o >>= 1) { const uint64_t x = (uint64_t)__shfl_xor((unsigned long long)v, o, 64); v = x > v ? x : v; }
        return v;
    };
    const uint64_t nw64 = (nwin + 63) >> 6;                            // a wave looks at the flags of 64 windows at a time
    for (uint64_t wb = (uint64_t)blockIdx.x * 4 + wv; wb < nw64; wb += (uint64_t)gridDim.x * 4) {
        const uint64_t w = wb * 64 + lane;
        const uint32_t fl = w < nwin ? wflag[w] : 0u;
        uint64_t pending = __builtin_amdgcn_ballot_w64(fl != 0);
        while (pending) {                                              // wave-uniform loop over the flagged windows
            const int src = __builtin_ctzll(pending);
            pending &= pending - 1;
            const uint32_t f = (uint32_t)__shfl((int)fl, src, 64);
            const uint64_t i0 = ((wb * 64 + (uint64_t)src) << 6) + (f - 1);      // first key of the long group
            const uint64_t grp = keys[i0] >> fmt.dbits;
            uint64_t b1 = 0, b2 = 0, carry_key = 0, carry_cnt = 0;
            bool carry = false;
            for (uint64_t pos = i0;; pos += 64) {
                const uint64_t idx = pos + lane;
                const uint64_t k = idx < n ? keys[idx] : ~0ull;
                const bool in = idx < n && (k >> fmt.dbits) == grp;
                const uint64_t inmask = __builtin_amdgcn_ballot_w64(in);
                const int nin = inmask == ~0ull ? 64 : __builtin_ctzll(~inmask);          // the group's keys are contiguous: leading lanes
                if (nin == 0) break;
                uint64_t kp = (uint64_t)__shfl_up((unsigned long long)k, 1, 64);
                if (lane == 0) kp = carry ? carry_key : ~k;
                const bool rs = lane < nin && kp != k;
                const uint64_t starts = __builtin_amdgcn_ballot_w64(rs);
                const int lead = starts ? __builtin_ctzll(starts) : nin;                  // keys that continue the carried run
                if (carry) {
                    carry_cnt += (uint64_t)lead;
                    if (lead < nin || nin < 64) {                                          // the carried run ends in this window
                        const uint64_t d = carry_key & dmask, a = (carry_cnt << 32) | (0xFFFFFFFFull - d), c2 = (carry_cnt << 32) | d;
                        b1 = a > b1 ? a : b1; b2 = c2 > b2 ? c2 : b2;
                        carry = false;
                    }
                }
                const uint64_t above = lane == 63 ? 0ull : (~0ull << (lane + 1));
                uint64_t v1 = 0, v2 = 0;
                bool open = false;                                                         // this lane's run reaches the window's end inside the group
                if (rs) {
                    const uint64_t nx = starts & above;
                    const int end = nx ? __builtin_ctzll(nx) : nin;
                    const uint64_t cnt = (uint64_t)(end - lane), d = k & dmask;
                    open = !nx && nin == 64;
                    if (!open) { v1 = (cnt << 32) | (0xFFFFFFFFull - d); v2 = (cnt << 32) | d; }
                }
                v1 = wmax(v1); v2 = wmax(v2);
                b1 = v1 > b1 ? v1 : b1; b2 = v2 > b2 ? v2 : b2;
                const uint64_t om = __builtin_amdgcn_ballot_w64(open);
                if (om) {                                                                  // the new carried run (at most one)
                    const int ol = __builtin_ctzll(om);
                    carry_key = (uint64_t)__shfl((unsigned long long)k, ol, 64);
                    carry_cnt = (uint64_t)(64 - ol);
                    carry = true;
                }
                if (nin < 64) break;
            }
            if (carry) {                                                                   // the group ended exactly at a window boundary
                const uint64_t d = carry_key & dmask, a = (carry_cnt << 32) | (0xFFFFFFFFull - d), c2 = (carry_cnt << 32) | d;
                b1 = a > b1 ? a : b1; b2 = c2 > b2 ? c2 : b2;
            }
            if ((int)(b1 >> 32) >= min_hits) {                                             // wave-uniform
                const uint32_t gq = qbegin + (uint32_t)(grp >> fmt.tbits), gt = (uint32_t)(grp & tmask);
                const bool mir = gq >= mirror_q0;
                if (lane == 0) {
                    s_q[wv][staged] = gq; s_t[wv][staged] = gt; s_d[wv][staged] = (int)(0xFFFFFFFFull - (b1 & 0xFFFFFFFFull)) - fmt.dbias;
                    if (mir) { s_q[wv][staged + 1] = gt; s_t[wv][staged + 1] = gq; s_d[wv][staged + 1] = -((int)(b2 & 0xFFFFFFFFull) - fmt.dbias); }
                }
                staged += mir ? 2u : 1u;
                __builtin_amdgcn_wave_barrier();
                if (staged >= 64) drain();
            }
        }
    }
    if (staged) drain();
}

// E4 sort key: [ query : 32 | 255 - score : 8 | target : 24 ]; score < min -> all-ones (sorted last)
__global__ void __launch_bounds__(256) select_key_kernel(uint64_t n, const uint32_t *cq, const uint32_t *ct, const int32_t *score,
                                                         int min_score, uint64_t *key, unsigned long long *n_kept) {
    unsigned long long kept = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const int s = score[i];
        const bool ok = s >= min_score;
        key[i] = ok ? ((uint64_t)cq[i] << 32) | ((uint64_t)(255 - s) << 24) | ct[i] : ~0ull;
        kept += ok;
    }
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_down(kept, o, 64);
    if ((threadIdx.x & 63) == 0 && kept) atomicAdd(n_kept, kept);
}

// E4 truncation: entry i of the (query, score desc, target) sorted list survives iff its rank inside its
// query's run is < max_seqs
__global__ void __launch_bounds__(256) rank_flag_kernel(const uint64_t *skey, uint64_t n, uint32_t max_seqs, uint32_t *flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        // rank inside the query's run < max_seqs  <=>  the entry max_seqs places earlier belongs to another (smaller) query: ONE load (r04; a binary
        // search for the run's first entry cost ~30 dependent loads per entry: 315 GB of fetches per configs[2] pass)
        const uint64_t qk = skey[i] & 0xFFFFFFFF00000000ull;
        flag[i] = (i < max_seqs || skey[i - max_seqs] < qk) ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(256) hit_scatter_kernel(const uint64_t *skey, const int32_t *sdiag, uint64_t n, const uint32_t *flag,
                                                          const uint64_t *pos, uint32_t *hq, uint32_t *ht, int32_t *hs, int32_t *hd) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint64_t k = skey[i], w = pos[i];
        hq[w] = (uint32_t)(k >> 32);
        ht[w] = (uint32_t)(k & 0xFFFFFFu);
        hs[w] = 255 - (int32_t)((k >> 24) & 0xFF);
        hd[w] = sdiag[i];
    }
}

// per-query hit counts from the query-sorted hit array
__global__ void __launch_bounds__(256) hit_count_kernel(const uint32_t *hq, uint64_t n_hits, uint32_t n, uint32_t *cnt) {
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) {
        uint64_t lo = 0, hi = n_hits;
        while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (hq[m] < q) lo = m + 1; else hi = m; }
        const uint64_t b = lo;
        hi = n_hits;
        while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (hq[m] <= q) lo = m + 1; else hi = m; }
        cnt[q] = (uint32_t)(lo - b);
    }
}

// ---- multi-GPU: merge of the all-gathered shard lists + pair ownership, on the device ----
// owner of the unordered pair {a,b}: mutual hits (q,t)/(t,q) must meet on one rank to share their DP.  The owner is a
// hash of the pair's REPRESENTATIVE QUERY (the shorter sequence, ties: smaller id - the orientation uc_align.hip computes),
// so a rank owns whole queries of the forward pass: its workgroup tasks stay as large as on one GPU (a hash of the pair
// itself scattered every query's pairs over all ranks: 8 x smaller tasks, the gapped stage scaled 3.7x on 8 ranks, not 8x)
__device__ __forceinline__ uint32_t pair_owner(uint32_t a, uint32_t b, const uint32_t *len, uint32_t world) {
    const uint32_t la = len[a], lb = len[b];
    const uint32_t rep = (la < lb || (la == lb && a < b)) ? a : b;
    uint32_t h = rep * 0x9E3779B1u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h % world;
}
__global__ void __launch_bounds__(256) merge_key_kernel(uint64_t n, const uint32_t *q, const uint32_t *t, const int32_t *score,
                                                        uint32_t nseq, uint64_t *key, uint32_t *bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const int s = score[i];
        if (q[i] >= nseq || t[i] >= nseq || s < 0 || s > 255) atomicAdd(bad, 1u);
        key[i] = ((uint64_t)q[i] << 32) | ((uint64_t)(255 - (s & 255)) << 24) | (t[i] & 0xFFFFFFu);
    }
}
__global__ void __launch_bounds__(256) owner_flag_kernel(const uint64_t *skey, uint64_t n, const uint32_t *len, uint32_t rank, uint32_t world, uint32_t *flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        if (flag[i] && pair_owner((uint32_t)(skey[i] >> 32), (uint32_t)(skey[i] & 0xFFFFFFu), len, world) != rank) flag[i] = 0;
}

// owner rank per installed hit + identity index (input of the stable partition by owner)
__global__ void __launch_bounds__(256) owner_key_kernel(uint64_t n, const uint32_t *hq, const uint32_t *ht, const uint32_t *len, uint32_t world,
                                                        uint32_t *okey, uint32_t *idx, unsigned long long *count /* world */) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t o = pair_owner(hq[i], ht[i], len, world);
        okey[i] = o;
        idx[i] = (uint32_t)i;
        // one atomic per wave and owner: the lanes of a wave that share an owner elect a leader
        for (uint32_t w = 0; w < world; w++) {
            const unsigned long long m = __ballot(o == w);
            if (m && (threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1) atomicAdd(count + w, (unsigned long long)__popcll(m));
        }
    }
}
__global__ void __launch_bounds__(256) owner_gather_kernel(uint64_t n, const uint32_t *idx, const uint32_t *hq, const uint32_t *ht, const int32_t *hs,
                                                           const int32_t *hd, uint32_t *oq, uint32_t *ot, int32_t *os, int32_t *od) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t j = idx[i];
        oq[i] = hq[j]; ot[i] = ht[j]; os[i] = hs[j]; od[i] = hd[j];
    }
}

static inline dim3 grid_for(uint64_t n, uint32_t cap = 16384) {
    const uint64_t b = (n + 255) / 256;
    return dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(b, cap)));
}

struct WidenU32 {
    __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; }
};

// work buffers of the prefilter, kept by the engine between calls (hipMalloc / hipFree of multi-GB buffers on every
// call cost tens of ms per step and, now and then, seconds)
struct PrefilterScratch {
    DevBuf<unsigned long long> d_counters;
    DevBuf<char> d_temp;
    DevBuf<uint32_t> d_koff, k_in, k_out, d_ent32, v_in32, d_okey, d_okey2, d_oidx, d_order;
    DevBuf<uint64_t> d_ent, v_in;
    DevBuf<uint32_t> d_cnt, d_flag, d_cq, d_ct, d_rpidx2, d_qsurv;
    DevBuf<uint64_t> d_keys, d_keys2, d_pos, d_skey, d_skey2, d_rval2, d_qbase, d_soff, d_qr;
    DevBuf<int32_t> d_cd, d_cd2, d_score, d_mval;      // d_mkey / d_mval: output of the sorted-run merge (merge_hits_dev)
    DevBuf<uint64_t> d_mkey;
    DevBuf<uint32_t> acc_q, acc_t, pass_q, pass_t;     // chunk loop of prefilter_impl: running top-M accumulator and the pass's lists (swapped in and out of the engine)
    DevBuf<int32_t> acc_s, acc_d, pass_s, pass_d;
    // distinct-k-mer enumeration (E2)
    DevBuf<uint8_t> d_kflag, d_wflag;
    DevBuf<uint32_t> d_qk, d_kid, d_dk, d_nsimk, d_drk, d_drk2, d_roff, d_nr, d_src, d_ph;
    DevBuf<uint64_t> d_drv, d_drv2, d_cumh, d_cumr, d_qh, d_qrn;
    DevBuf<RankRec> d_rec;
    DevBuf<uint32_t> d_kbits;       // presence bitmap of the chunk's k-mers
    // every buffer (for the size bookkeeping below)
    template <class F> void each(F f) {
        f(d_counters); f(d_temp); f(d_koff); f(k_in); f(k_out); f(d_ent32); f(v_in32); f(d_okey); f(d_okey2); f(d_oidx); f(d_order);
        f(d_ent); f(v_in); f(d_cnt); f(d_flag); f(d_cq); f(d_ct); f(d_rpidx2); f(d_qsurv); f(d_keys); f(d_keys2); f(d_pos); f(d_skey);
        f(d_skey2); f(d_rval2); f(d_qbase); f(d_soff); f(d_qr); f(d_cd); f(d_cd2); f(d_score); f(d_kflag); f(d_wflag); f(d_qk); f(d_kid); f(d_dk);
        f(d_nsimk); f(d_drk); f(d_drk2); f(d_roff); f(d_nr); f(d_src); f(d_ph); f(d_drv); f(d_drv2); f(d_cumh); f(d_cumr); f(d_qh); f(d_qrn); f(d_rec); f(d_kbits); f(d_mkey); f(d_mval); f(acc_q); f(acc_t); f(pass_q); f(pass_t); f(acc_s); f(acc_d); f(pass_s); f(pass_d);
    }
    size_t bytes() {
        size_t b = 0;
        each([&](auto &x) { b += x.cap * sizeof(*x.p); });
        return b;
    }
    // Large databases grow this set to well over 100 GB (per-position arrays, run lists, 30 GiB of key regions); the gapped
    // stage sizes its own batches by the memory that is FREE, so above `limit` the big buffers go back before it starts
    // (re-allocating them costs milliseconds per step at a scale where a step takes a minute; small databases keep everything).
    void release_bytes(size_t target) {      // largest buffers first until `target` bytes are back (target >= bytes(): everything of 1 GiB and more, the r01-r05 trim)
        if (getenv("UC_TIMING")) {
            fprintf(stderr, "unicore-cluster[timing]: prefilter scratch %.1f GiB, %.1f GiB to give back before the gapped stage; buffers >= 2 GiB (index in PrefilterScratch::each: GiB):", bytes() / 1073741824.0, target / 1073741824.0);
            int i = 0;
            each([&](auto &x) { const double g = x.cap * sizeof(*x.p) / 1073741824.0; if (g >= 2.0) fprintf(stderr, " %d:%.1f", i, g); i++; });
            fprintf(stderr, "\n");
        }
        if (!target) return;
        if (target >= bytes()) { each([&](auto &x) { if (x.cap * sizeof(*x.p) >= ((size_t)1 << 30)) x.release(); }); return; }
        size_t freed = 0;
        while (freed < target) {
            size_t best = 0;
            each([&](auto &x) { best = std::max(best, x.cap * sizeof(*x.p)); });
            if (best < ((size_t)256 << 20)) break;
            bool done = false;
            each([&](auto &x) { if (!done && x.cap * sizeof(*x.p) == best) { x.release(); done = true; } });
            freed += best;
        }
    }
};
void free_prefilter_scratch(PrefilterScratch *p) { delete p; }

// A destroyed engine parks its work buffers here (one set per device) and the next engine of the process takes them
// over: freeing and re-allocating tens of GB between two uc_cluster calls is usually free, but now and then the next
// hipMalloc then takes 1-3 s (measured: 1 in ~5 calls).  UC_KEEP_SCRATCH=0 releases them with the engine instead.
namespace {
std::mutex g_park_mutex;
PrefilterScratch *g_parked_pre[16] = {};
bool keep_scratch() { const char *e = getenv("UC_KEEP_SCRATCH"); return !(e && e[0] == '0'); }
}  // namespace
void park_prefilter_scratch(PrefilterScratch *p, int device) {
    if (!p) return;
    if (keep_scratch() && device >= 0 && device < 16) {
        std::lock_guard<std::mutex> g(g_park_mutex);
        if (!g_parked_pre[device]) { g_parked_pre[device] = p; return; }
    }
    delete p;
}
PrefilterScratch *take_parked_prefilter_scratch(int device) {   // nullptr if nothing is parked (uc_release_scratch)
    if (device < 0 || device >= 16) return nullptr;
    std::lock_guard<std::mutex> g(g_park_mutex);
    PrefilterScratch *p = g_parked_pre[device];
    g_parked_pre[device] = nullptr;
    return p;
}
PrefilterScratch *take_prefilter_scratch(int device) {
    if (device >= 0 && device < 16) {
        std::lock_guard<std::mutex> g(g_park_mutex);
        if (PrefilterScratch *p = g_parked_pre[device]) { g_parked_pre[device] = nullptr; return p; }
    }
    return new PrefilterScratch;
}

// the prefilter's work buffers stay allocated between steps (re-allocating the 34 GiB key regions alone costs a second per
// step at configs[1]) unless they hold more than 40 % of the device memory: then the gapped stage, which sizes its batches by
// the free memory, gets them back (2.5 M sequences: 170 GB)
// r06: the 40 % rule only where the gapped stage really sizes something by the free memory - the traceback-byte matrices of --min-seq-id / search (MODE 7).
// Without them the gapped stage needs ~150 B per pair of a 256 M-pair batch and nothing else: the prefilter's buffers stay up to 65 % of the device, and if
// that is ever too much the stage's out-of-memory handler gives them back (Engine::relieve_pressure).  Why it matters: releasing ~100 GB after every pass
// means allocating them again in the next one, and on a box whose device memory has not been touched since it came up that costs seconds PER PASS for as long
// as untouched memory is left (configs[2], first process on such a box: prefilter 9.2 s instead of 5.3 s per pass - profiles/r06/c3_first_process.txt).
// -> bytes of the prefilter's buffers to give back before the gapped stage starts (0 = keep everything)
static size_t scratch_release_target(bool gapped_stage_sizes_by_free_memory, uint64_t n_hits, size_t scratch_bytes, bool gapped_scratch_warm) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) return scratch_bytes > ((size_t)96 << 30) ? scratch_bytes : 0;
    if (gapped_stage_sizes_by_free_memory) return scratch_bytes > (size_t)((double)tot * 0.4) ? scratch_bytes : 0;       // the 40 % rule of r01-r05: everything >= 1 GiB
    // ... otherwise only what the gapped stage will not fit beside: its plans and pass arrays take up to ~600 B per pair of a 256 M-pair batch (measured at
    // configs[2]: it ran out of memory beside 133 GiB of kept buffers with 126 GiB free) + 44 B of records per listed pair.  Kept buffers that the stage's
    // out-of-memory handler has to take back after all cost more than releasing them here: the block allocated right behind a 130 GiB release waits seconds
    // for it (configs[2]: 41.4 s per pass instead of 33.3, profiles/r06/c3_memory_edge.txt).  The largest buffers go first, only as many as needed.
    // (warm: the gapped stage still holds the buffers of a call at least this large - repeated steps, the shrinking rounds of a cascade: only the margin)
    const size_t need = gapped_scratch_warm ? ((size_t)8 << 30)
                                            : (size_t)600 * (size_t)std::min<uint64_t>(n_hits, 256ull << 20) + (size_t)44 * (size_t)n_hits + ((size_t)8 << 30);
    return fr >= need ? 0 : std::min(scratch_bytes, need - fr);
}

// E1-E4 for a target range.  Large ranges are processed as several index chunks whose per-query top-M lists are
// merged on the device (lossless, same argument as the multi-GPU shards): the double-hit filter keeps one query's
// (target, diagonal) hashes in 2 x 2^19 LDS bits, which only works while a query has well under ~500 k k-mer hits,
// i.e. up to ~100 M target residues per chunk at default sensitivity (at 760 M residues in one chunk 75 % of the
// hits survived the filter and the key sort took 2/3 of the run).
void Engine::prefilter(uint32_t tbegin, uint32_t tend, uint32_t qbegin, uint32_t qend) { prefilter_impl(tbegin, tend, qbegin, qend, false); }

// A rank of a symmetric N-rank pass: the shard's own block (with its inner triangle if it spans several chunks) and one mirrored block per other
// query range; the lists of the blocks (disjoint candidate sets, each truncated to its own top-M) are concatenated and merged once.
void Engine::prefilter_cells(uint32_t tbegin, uint32_t tend, const std::vector<std::pair<uint32_t, uint32_t>> &others) {
    PressureScope ps(*this, 0);
    DevBuf<uint32_t> cq, ct;
    DevBuf<int32_t> cs, cd;
    uint64_t cat_n = 0;
    auto take = [&]() {       // append the engine's current lists (grouped or not) to the accumulator
        if (!n_hits) return;
        const uint64_t tot = cat_n + n_hits;
        cq.grow_preserve(tot, cat_n, stream); ct.grow_preserve(tot, cat_n, stream); cs.grow_preserve(tot, cat_n, stream); cd.grow_preserve(tot, cat_n, stream);
        UC_HIP(hipMemcpyAsync(cq.p + cat_n, d_hq.p, n_hits * 4, hipMemcpyDeviceToDevice, stream));
        UC_HIP(hipMemcpyAsync(ct.p + cat_n, d_ht.p, n_hits * 4, hipMemcpyDeviceToDevice, stream));
        UC_HIP(hipMemcpyAsync(cs.p + cat_n, d_hs.p, n_hits * 4, hipMemcpyDeviceToDevice, stream));
        UC_HIP(hipMemcpyAsync(cd.p + cat_n, d_hd.p, n_hits * 4, hipMemcpyDeviceToDevice, stream));
        UC_HIP(hipStreamSynchronize(stream));
        cat_n = tot;
    };
    const uint64_t before = stats.n_prefilter_hits;
    prefilter_impl(tbegin, tend, tbegin, tend, false);
    take();
    for (const auto &r : others) {
        if (r.first >= r.second) continue;
        if (r.first < tend && r.second > tbegin) fail(UC_ERR_GENERIC, "prefilter_cells: a mirrored query range overlaps the target shard");
        prefilter_impl(tbegin, tend, r.first, r.second, true);
        take();
    }
    const uint64_t kept = import_hits_dev(cat_n, cq.p, ct.p, cs.p, cd.p, 0, 1);
    stats.n_prefilter_hits = before + kept;
    if (pre) pre->release_bytes(scratch_release_target(p.min_seq_id > 0.0f || p.want_tb, n_hits, pre->bytes(), aln != nullptr && (double)last_align_hits >= 0.9 * (double)n_hits));
}

void Engine::prefilter_impl(uint32_t tbegin, uint32_t tend, uint32_t qbegin, uint32_t qend, bool mirror_all) {
    PressureScope ps(*this, 0);
    if (!have_db) fail(UC_ERR_ARGS, "no database loaded");
    if (tbegin > tend || tend > hdb.n) fail(UC_ERR_ARGS, "prefilter: bad target range");
    if (qend == UINT32_MAX) qend = hdb.n;
    if (qbegin > qend || qend > hdb.n) fail(UC_ERR_ARGS, "prefilter: bad query range");
    uint64_t chunk_res = prefilter_chunk_residues;
    if (const char *ev = getenv("UC_PREFILTER_CHUNK_RES")) chunk_res = std::max<uint64_t>(1, strtoull(ev, nullptr, 10));
    // The chunk size that keeps a query's hits inside the LDS filter depends on how many k-mer hits a target residue
    // attracts, i.e. on the sensitivity (-s 7.5 gives ~30x the hits of the default 4): prefilter_one measures the density
    // (k-mer hits per query residue) of its first batch and gives up before expanding anything if the chunk is too dense;
    // the range is then re-cut into proportionally smaller chunks (one wasted enumeration pass).
    const double DENSITY_LIMIT = 400.0;   // hits per query residue and chunk (C2 whole DB at -s 4: 124)
    for (int attempt = 0;; attempt++) {
        std::vector<std::pair<uint32_t, uint32_t>> chunks;
        for (uint32_t b = tbegin; b < tend;) {
            uint32_t e = b;
            uint64_t res = 0;
            while (e < tend && (e == b || res + h_len[e] <= chunk_res)) res += h_len[e++];
            chunks.emplace_back(b, e);
            b = e;
        }
        double density = 0;
        const double limit = attempt < 6 ? DENSITY_LIMIT : 0.0;      // after 6 re-cuts: run with what we have
        bool ok = true;
        // All-vs-all over several chunks under a symmetric matrix: the k-mer hit relation is symmetric, so the chunk x chunk grid needs only its
        // upper triangle.  The pass over chunk c matches the queries FROM chunk c onwards; the pairs (query behind the chunk, target in it) also
        // yield the candidates of the pairs the other way round (diag_select_kernel), i.e. what the skipped passes (query in c, target chunk
        // behind it) would have found: ~(1 + 1/C) / 2 of the k-mer hits are expanded.  The lists of a pass come back ungrouped and are merged
        // like the chunk lists always were (lossless: per-pass top-M of disjoint candidate sets).  UC_PREFILTER_SYMMETRIC=0: every pass matches all queries.
        const bool triangle = chunks.size() > 1 && tbegin == qbegin && tend == qend && p.mat_symmetric &&
                              !(getenv("UC_PREFILTER_SYMMETRIC") && atoi(getenv("UC_PREFILTER_SYMMETRIC")) == 0);
        if (chunks.size() <= 1) {
            // (mirror_all — prefilter_cells only: every query lies outside the shard and yields the pair the other way round as well; the lists come
            // back ungrouped and are merged by the caller)
            ok = prefilter_one(tbegin, tend, qbegin, qend, true, limit, &density, mirror_all ? qbegin : UINT32_MAX);
            if (ok) { stats.n_prefilter_hits += n_hits; if (pre && !mirror_all) pre->release_bytes(scratch_release_target(p.min_seq_id > 0.0f || p.want_tb, n_hits, pre->bytes(), aln != nullptr && (double)last_align_hits >= 0.9 * (double)n_hits)); return; }
        } else {
            // (Measured and reverted in r04: concatenating the per-pass lists and merging ONCE at the end — 1.5-2 G records in one sort instead of a
            // running top-M accumulator of <= max_seqs x queries — made configs[2] SLOWER, 35.7 -> 39.3 s per pass: the single merge needs ~64 B per
            // record of work buffers at the moment the key regions are largest.  The accumulator is merged after every pass.)
            // (r06: the accumulator / pass arrays live in the scratch set - up to 4 x 7 GB at configs[2] were allocated and freed by every call)
            if (!pre) pre = take_prefilter_scratch(device);
            DevBuf<uint32_t> &aq = pre->acc_q, &at = pre->acc_t, &tq = pre->pass_q, &tt = pre->pass_t;
            DevBuf<int32_t> &as = pre->acc_s, &ad = pre->acc_d, &ts = pre->pass_s, &td = pre->pass_d;
            uint64_t acc_n = 0;
            bool installed = false;       // the last pass's merge leaves its result installed in the engine: no second merge of the accumulator
            for (size_t c = 0; c < chunks.size() && ok; c++) {
                const bool mir = mirror_all || (triangle && c + 1 < chunks.size());       // (the last chunk has no queries behind it: a plain pass over its own queries)
                ok = prefilter_one(chunks[c].first, chunks[c].second, triangle ? chunks[c].first : qbegin, qend, c == 0, c == 0 ? limit : 0.0, &density,
                                   mirror_all ? qbegin : mir ? chunks[c].second : UINT32_MAX);
                if (!ok) break;
                if (!mir && (c == 0 || acc_n == 0)) {
                    aq.swap(d_hq); at.swap(d_ht); as.swap(d_hs); ad.swap(d_hd);
                    acc_n = n_hits;
                } else if (n_hits) {       // (a mirrored pass always comes here: its lists are ungrouped)
                    // the pass's lists leave the engine (swap, no copy), the merge installs accumulator + pass in their place.  The accumulator is in key
                    // order, a plain pass's lists are too: only a mirrored pass's records are sorted before the two runs are merged (merge_hits_dev)
                    const uint64_t n_pass = n_hits;
                    tq.swap(d_hq); tt.swap(d_ht); ts.swap(d_hs); td.swap(d_hd);
                    acc_n = merge_hits_dev(acc_n, aq.p, at.p, as.p, ad.p, n_pass, tq.p, tt.p, ts.p, td.p, /*sorted2=*/!mir, 0, 1);      // merge + truncate to max_seqs
                    if (c + 1 == chunks.size()) installed = true;
                    else { aq.swap(d_hq); at.swap(d_ht); as.swap(d_hs); ad.swap(d_hd); }
                }
            }
            if (ok) {
                // install the accumulated lists (also rebuilds the per-query counts) unless the last merge already did
                if (!installed) merge_hits_dev(acc_n, aq.p, at.p, as.p, ad.p, 0, nullptr, nullptr, nullptr, nullptr, true, 0, 1);     // (in key order already: counts only)
                stats.n_prefilter_hits += n_hits;
                if (pre) pre->release_bytes(scratch_release_target(p.min_seq_id > 0.0f || p.want_tb, n_hits, pre->bytes(), aln != nullptr && (double)last_align_hits >= 0.9 * (double)n_hits));
                return;
            }
        }
        // too dense: smaller chunks, proportionally (and a little more)
        const uint64_t cur = std::min<uint64_t>(chunk_res, std::max<uint64_t>(1, (uint64_t)h_poff[chunks[0].second] - h_poff[chunks[0].first]));
        chunk_res = std::max<uint64_t>(1u << 16, (uint64_t)((double)cur * DENSITY_LIMIT / density * 0.75));
        logf(3, "unicore-cluster: prefilter: %.0f k-mer hits per query residue in a chunk of %llu residues; re-cutting the targets into chunks of %llu\n",
             density, (unsigned long long)cur, (unsigned long long)chunk_res);
    }
}

// returns false (nothing installed) if density_limit > 0 and the first query batch exceeds it; *density_out = k-mer hits
// per query residue of that batch
bool Engine::prefilter_one(uint32_t tbegin, uint32_t tend, uint32_t qbegin, uint32_t qend, bool count_sims, double density_limit,
                           double *density_out, uint32_t mirror_q0) {
    const bool mirrored = mirror_q0 != UINT32_MAX;    // symmetric pass (diag_select_kernel): the lists leave this function ungrouped, the caller merges them
    UC_HIP(hipSetDevice(device));
    const uint32_t n = hdb.n;
    KmerCfg cfg;
    for (int m = 0; m < K; m++) cfg.koff[m] = p.koff[m];
    cfg.span = p.span;
    cfg.thr = p.kmer_thr;

    hit_cnt.assign(n, 0);
    hit_off.assign((size_t)n + 1, 0);
    n_hits = 0;
    alns_valid = false;
    clear_edges();
    const uint64_t sims_before = stats.n_sim_kmers;   // stats accumulate over calls: the byte count below needs THIS call's share

    if (n > (1u << 24)) fail(UC_ERR_GENERIC, "prefilter: %u sequences exceed the 2^24 limit of the hit keys", n);
    // counters: [0] similar k-mers, [1] kept candidates, [2] ungapped overlap residues, [3] run cursor,
    //           [4] k-mer hits of the batch, [5] key cursor, [6] candidate cursor
    if (!pre) pre = take_prefilter_scratch(device);
    PrefilterScratch &S = *pre;
    DevBuf<unsigned long long> &d_counters = S.d_counters;
    d_counters.reserve(8);
    UC_HIP(hipMemsetAsync(d_counters.p, 0, 64, stream));
    DevBuf<char> &d_temp = S.d_temp;
    auto temp_reserve = [&](size_t bytes) { d_temp.reserve(bytes + 256); };

    // ------------------------------------------------------------ E1: index of targets [tbegin, tend)
    Timer t_index;
    timed_ms_begin();
    const uint32_t tp0 = h_poff[tbegin], tp1 = h_poff[tend];
    const uint32_t nres = tp1 - tp0;
    DevBuf<uint32_t> &d_koff = S.d_koff;
    DevBuf<uint64_t> &d_ent = S.d_ent;      // wide index entries sorted by k-mer: [sequence : 32 | position : 16]
    DevBuf<uint32_t> &d_ent32 = S.d_ent32;  // compact index entries: [sequence - tbegin | position : dbits - 1]
    d_koff.reserve((size_t)KSPACE + 1);
    KeyFmt fmt;
    {
        const uint32_t m = std::min<uint32_t>(std::max<uint32_t>(max_len, 2), 65536u);
        fmt.dbits = 1;
        while ((1u << (fmt.dbits - 1)) < m) fmt.dbits++;
        fmt.dbias = 1 << (fmt.dbits - 1);
        fmt.tbits = 1;
        while ((1u << fmt.tbits) < n) fmt.tbits++;
        int rbits = 1;                                   // bits of a target id relative to this chunk
        while ((1ull << rbits) < (uint64_t)std::max<uint32_t>(tend - tbegin, 1)) rbits++;
        fmt.compact = rbits + fmt.dbits <= 32 && !getenv("UC_PREFILTER_WIDE");
        fmt.tbase = tbegin;
    }
    const void *ent_p = nullptr;
    uint32_t n_entries = 0;
    {
        DevBuf<uint32_t> &k_in = S.k_in, &k_out = S.k_out, &v_in32 = S.v_in32;
        DevBuf<uint64_t> &v_in = S.v_in;
        const size_t cap = std::max<uint32_t>(nres, 1);
        k_in.reserve(cap); k_out.reserve(cap);
        if (fmt.compact) { v_in32.reserve(cap); d_ent32.reserve(cap); ent_p = d_ent32.p; }
        else { v_in.reserve(cap); d_ent.reserve(cap); ent_p = d_ent.p; }
        if (nres) {
            hipLaunchKernelGGL(kmer_extract_kernel, grid_for(nres), dim3(256), 0, stream, ddb, tbegin, tend, cfg, tp0, tp1, k_in.p,
                               fmt.compact ? nullptr : v_in.p, fmt.compact ? v_in32.p : nullptr, fmt.dbits - 1);
            size_t tb = 0;
            if (fmt.compact) {
                UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, k_in.p, k_out.p, v_in32.p, d_ent32.p, (size_t)nres, 0u, 32u, stream));
                temp_reserve(tb);
                UC_HIP(rocprim::radix_sort_pairs(d_temp.p, tb, k_in.p, k_out.p, v_in32.p, d_ent32.p, (size_t)nres, 0u, 32u, stream));
            } else {
                UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, k_in.p, k_out.p, v_in.p, d_ent.p, (size_t)nres, 0u, 32u, stream));
                temp_reserve(tb);
                UC_HIP(rocprim::radix_sort_pairs(d_temp.p, tb, k_in.p, k_out.p, v_in.p, d_ent.p, (size_t)nres, 0u, 32u, stream));
            }
        }
        // number of valid entries = first index with key >= KSPACE: the offsets kernel's last slot
        hipLaunchKernelGGL(kmer_offsets_kernel, grid_for((uint64_t)KSPACE + 1), dim3(256), 0, stream, k_out.p, nres, d_koff.p);
        S.d_kbits.reserve((KSPACE + 31) / 32);
        hipLaunchKernelGGL(kmer_bits_kernel, grid_for((KSPACE + 31) / 32), dim3(256), 0, stream, (const uint32_t *)d_koff.p, S.d_kbits.p);
        UC_HIP(hipMemcpyAsync(&n_entries, d_koff.p + KSPACE, 4, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipStreamSynchronize(stream));
    }
    UC_HIP(hipGetLastError());
    double gpu_ms = timed_ms_end();
    stats.n_index_entries += n_entries;
    stats.algorithmic_bytes[UC_ST_INDEX] += 6ull * n_entries + 8ull * KSPACE;
    stats.stage_seconds[UC_ST_INDEX] += t_index.seconds();

    // ------------------------------------------------------------ E2-E4 over query batches
    // keys per batch (the filter's regions hold < 2^32 keys).  1.5 G keys = 12 GiB of regions: a one-shot `foldseek cluster` process
    // pays for every byte it allocates (34 GiB of regions at 3.75 G keys: 5.2 s from process start to clust.tsv at configs[1]
    // instead of 1.25 s), and a resident engine loses nothing measurable (681 vs 681 ms per step; UC_HIT_CAP overrides)
    const uint64_t HIT_CAP = getenv("UC_HIT_CAP") ? std::max<uint64_t>(1u << 20, strtoull(getenv("UC_HIT_CAP"), nullptr, 10)) : (3ull << 29);
    // ... unless the chunk has so many hits that the batches would run into the hundreds (2.5 M sequences: 1.5e12 hits): then the
    // per-batch costs outweigh the allocation and the regions take 3.75 G keys (30 GiB)
    const uint64_t HIT_CAP_BIG = getenv("UC_HIT_CAP") ? HIT_CAP : (15ull << 28);
    uint64_t hit_cap = HIT_CAP;
    const uint64_t RUN_MAX = 1ull << 29;       // runs per batch (6 GiB + 6 GiB sort double buffer)
    uint64_t run_cap = 1ull << 20;
    DevBuf<uint32_t> &d_cnt = S.d_cnt, &d_flag = S.d_flag, &d_cq = S.d_cq, &d_ct = S.d_ct, &d_rpidx2 = S.d_rpidx2, &d_qsurv = S.d_qsurv;
    DevBuf<uint64_t> &d_keys = S.d_keys, &d_keys2 = S.d_keys2, &d_pos = S.d_pos, &d_skey = S.d_skey, &d_skey2 = S.d_skey2, &d_rval2 = S.d_rval2,
                     &d_qbase = S.d_qbase, &d_soff = S.d_soff;
    DevBuf<int32_t> &d_cd = S.d_cd, &d_cd2 = S.d_cd2, &d_score = S.d_score;
    uint64_t n_hits_total = 0, n_cand_total = 0, cand_cap = 0;
    double t_kmer = 0, t_ung = 0, t_sel = 0;
    const auto sim_grid = [](uint64_t items) {
        return dim3((uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (items + 4 * SIM_MIN_WAVE_POS - 1) / (4 * SIM_MIN_WAVE_POS)), SIM_MAX_BLOCKS));
    };

    // ---- distinct mode, once per target chunk and query "super-batch" [sa, sb) (normally all queries; cut when the run lists
    //      of its distinct k-mers exceed DRUN_MAX, i.e. at high sensitivity): similar k-mers and index ranges of every
    //      DISTINCT query k-mer, then per query position the length and source of its run list and its k-mer hits; exact
    //      per-query totals for the batch plan ----
    const uint32_t first_query = qbegin;
    const uint64_t DRUN_MAX = getenv("UC_DRUN_MAX") ? std::max<uint64_t>(1, strtoull(getenv("UC_DRUN_MAX"), nullptr, 10)) : (1ull << 31);   // 24 GiB + 24 GiB sort double buffer (env: tests)
    uint32_t P0 = 0, NP = 0;
    std::vector<uint64_t> h_qh, h_qr;          // k-mer hits / runs per query of the super-batch
    uint64_t over_runs = 0;
    // 0 = planned, 1 = density limit exceeded (first super-batch only), 2 = too many runs (over_runs says how many)
    const auto plan_superbatch = [&](uint32_t sa, uint32_t sb) -> int {
        uint64_t plan_sims = 0, plan_hits = 0;
        P0 = h_poff[sa]; NP = h_poff[sb] - P0;
        const uint32_t qbegin = sa, qend = sb;   // this block works on the super-batch only
        Timer t_p;
        timed_ms_begin();
        const uint32_t nqa = qend - qbegin;
        S.d_qk.reserve((size_t)NP + 1); S.d_kflag.reserve((size_t)KSPACE + 1); S.d_kid.reserve((size_t)KSPACE + 1);
        UC_HIP(hipMemsetAsync(S.d_kflag.p, 0, (size_t)KSPACE + 1, stream));
        hipLaunchKernelGGL(query_kmer_kernel, grid_for(NP), dim3(256), 0, stream, ddb, cfg, qbegin, qend, P0, P0 + NP, S.d_qk.p, S.d_kflag.p);
        size_t tb = 0;
        auto fin = rocprim::make_transform_iterator(S.d_kflag.p, FlagToU32());
        UC_HIP(rocprim::exclusive_scan(nullptr, tb, fin, S.d_kid.p, 0u, (size_t)KSPACE + 1, rocprim::plus<uint32_t>(), stream));
        temp_reserve(tb);
        UC_HIP(rocprim::exclusive_scan(d_temp.p, tb, fin, S.d_kid.p, 0u, (size_t)KSPACE + 1, rocprim::plus<uint32_t>(), stream));
        uint32_t nd = 0;
        UC_HIP(hipMemcpyAsync(&nd, S.d_kid.p + KSPACE, 4, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipStreamSynchronize(stream));
        S.d_dk.reserve(std::max<uint32_t>(nd, 1)); S.d_nsimk.reserve(std::max<uint32_t>(nd, 1));
        UC_HIP(hipMemsetAsync(S.d_nsimk.p, 0, (size_t)std::max<uint32_t>(nd, 1) * 4, stream));
        hipLaunchKernelGGL(distinct_kmer_kernel, grid_for(KSPACE), dim3(256), 0, stream, S.d_kflag.p, S.d_kid.p, S.d_dk.p);
        if (nqa > 1 && nd > (1u << 16)) {   // run-count estimate from every 64th distinct k-mer: cut the super-batch BEFORE the full enumeration
            const uint32_t st = 64;
            UC_HIP(hipMemsetAsync(d_counters.p + 3, 0, 32, stream));
            UC_HIP(hipMemsetAsync(d_counters.p, 0, 8, stream));
            hipLaunchKernelGGL(sim_runs_kernel, sim_grid((nd + st - 1) / st), dim3(256), 0, stream, ddb, cfg, 0u, 0u, 0u, nd, d_koff.p, RunList{nullptr, nullptr, 0},
                               d_counters.p, S.d_dk.p, (uint32_t *)nullptr, st, (const uint32_t *)S.d_kbits.p);
            unsigned long long c5[5];
            UC_HIP(hipMemcpyAsync(c5, d_counters.p, 40, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipStreamSynchronize(stream));
            if ((double)c5[3] * st > 0.9 * (double)DRUN_MAX) { over_runs = (uint64_t)((double)c5[3] * st / 0.9); gpu_ms += timed_ms_end(); t_kmer += t_p.seconds(); return 2; }
        }
        uint64_t n_druns = 0, drun_cap = S.d_drk.cap;
        for (;;) {   // runs of the distinct k-mers, tagged with the k-mer's rank
            drun_cap = std::min<uint64_t>(1ull << 32, std::max<uint64_t>(drun_cap, std::max<uint64_t>(1u << 20, std::min<uint64_t>(DRUN_MAX, (uint64_t)nd * 16))));
            S.d_drk.reserve(drun_cap); S.d_drv.reserve(drun_cap);
            UC_HIP(hipMemsetAsync(d_counters.p + 3, 0, 32, stream));
            UC_HIP(hipMemsetAsync(d_counters.p, 0, 8, stream));
            const RunList rl{S.d_drk.p, S.d_drv.p, drun_cap};
            hipLaunchKernelGGL(sim_runs_kernel, sim_grid(nd), dim3(256), 0, stream, ddb, cfg, 0u, 0u, 0u, nd, d_koff.p, rl, d_counters.p, S.d_dk.p, S.d_nsimk.p, 1u,
                               (const uint32_t *)S.d_kbits.p);
            unsigned long long c5[5];
            UC_HIP(hipMemcpyAsync(c5, d_counters.p, 40, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipStreamSynchronize(stream));
            n_druns = c5[3];
            if (n_druns > DRUN_MAX && nqa > 1) { over_runs = n_druns; gpu_ms += timed_ms_end(); t_kmer += t_p.seconds(); return 2; }
            if (n_druns > drun_cap) {
                if (n_druns >= (1ull << 32)) fail(UC_ERR_GENERIC, "%u distinct k-mers of query %u produce %llu index ranges", nd, qbegin, (unsigned long long)n_druns);
                drun_cap = n_druns;
                UC_HIP(hipMemsetAsync(S.d_nsimk.p, 0, (size_t)std::max<uint32_t>(nd, 1) * 4, stream));
                continue;
            }
            break;
        }
        unsigned dbits = 1;
        while ((1ull << dbits) < nd) dbits++;
        S.d_drk2.reserve(drun_cap); S.d_drv2.reserve(drun_cap);
        if (n_druns) {
            UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, S.d_drk.p, S.d_drk2.p, S.d_drv.p, S.d_drv2.p, (size_t)n_druns, 0u, dbits, stream));
            temp_reserve(tb);
            UC_HIP(rocprim::radix_sort_pairs(d_temp.p, tb, S.d_drk.p, S.d_drk2.p, S.d_drv.p, S.d_drv2.p, (size_t)n_druns, 0u, dbits, stream));
        }
        S.d_roff.reserve((size_t)nd + 1); S.d_rec.reserve(std::max<uint32_t>(nd, 1));
        hipLaunchKernelGGL(rank_offsets_kernel, grid_for((uint64_t)nd + 1), dim3(256), 0, stream, S.d_drk2.p, (uint32_t)n_druns, nd, S.d_roff.p);
        if (nd) hipLaunchKernelGGL(rank_rec_kernel, grid_for(nd), dim3(256), 0, stream, S.d_roff.p, S.d_drv2.p, S.d_nsimk.p, nd, S.d_rec.p);
        // per position (one trailing zero element so that the exclusive sums end with the totals)
        S.d_nr.reserve((size_t)NP + 1); S.d_src.reserve((size_t)NP + 1); S.d_ph.reserve((size_t)NP + 1);
        S.d_cumh.reserve((size_t)NP + 1); S.d_cumr.reserve((size_t)NP + 1);
        UC_HIP(hipMemsetAsync(d_counters.p, 0, 8, stream));
        UC_HIP(hipMemsetAsync(S.d_nr.p + NP, 0, 4, stream));
        UC_HIP(hipMemsetAsync(S.d_ph.p + NP, 0, 4, stream));
        hipLaunchKernelGGL(position_runs_kernel, grid_for(NP), dim3(256), 0, stream, S.d_qk.p, NP, S.d_kid.p, S.d_rec.p, S.d_nr.p, S.d_src.p, S.d_ph.p, d_counters.p);
        auto hin = rocprim::make_transform_iterator(S.d_ph.p, WidenU32());
        auto rin = rocprim::make_transform_iterator(S.d_nr.p, WidenU32());
        UC_HIP(rocprim::exclusive_scan(nullptr, tb, hin, S.d_cumh.p, (uint64_t)0, (size_t)NP + 1, rocprim::plus<uint64_t>(), stream));
        temp_reserve(tb);
        UC_HIP(rocprim::exclusive_scan(d_temp.p, tb, hin, S.d_cumh.p, (uint64_t)0, (size_t)NP + 1, rocprim::plus<uint64_t>(), stream));
        UC_HIP(rocprim::exclusive_scan(d_temp.p, tb, rin, S.d_cumr.p, (uint64_t)0, (size_t)NP + 1, rocprim::plus<uint64_t>(), stream));
        S.d_qh.reserve(nqa); S.d_qrn.reserve(nqa);
        hipLaunchKernelGGL(query_totals_kernel, grid_for(nqa), dim3(256), 0, stream, ddb.off, qbegin, nqa, P0, S.d_cumh.p, S.d_cumr.p, S.d_qh.p, S.d_qrn.p);
        h_qh.resize(nqa); h_qr.resize(nqa);
        unsigned long long c0 = 0;
        UC_HIP(hipMemcpyAsync(h_qh.data(), S.d_qh.p, (size_t)nqa * 8, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipMemcpyAsync(h_qr.data(), S.d_qrn.p, (size_t)nqa * 8, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipMemcpyAsync(&c0, d_counters.p, 8, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipStreamSynchronize(stream));
        plan_sims = c0;
        for (uint32_t i = 0; i < nqa; i++) plan_hits += h_qh[i];
        hit_cap = plan_hits > 16 * HIT_CAP ? HIT_CAP_BIG : HIT_CAP;
        gpu_ms += timed_ms_end();
        t_kmer += t_p.seconds();
        if (sa == first_query) {
            if (density_out) *density_out = (double)plan_hits / std::max<uint32_t>(1, NP);
            if (density_limit > 0 && p.min_diag_hits >= 2 && tend - tbegin > 1 && (double)plan_hits / std::max<uint32_t>(1, NP) > density_limit) return 1;
        }
        if (count_sims) stats.n_sim_kmers += plan_sims;
        return 0;
    };
    uint32_t sb_end = qbegin;                  // end of the planned super-batch
    double sb_frac = 1.0;                      // share of the remaining queries to try next

    uint32_t sb_begin = qbegin;
    for (uint32_t qa = qbegin; qa < qend;) {
        if (qa >= sb_end) {
            for (;;) {
                const uint32_t left = qend - qa;
                const uint32_t sb = sb_frac >= 1.0 ? qend : qa + std::max<uint32_t>(1, std::min<uint32_t>(left, (uint32_t)(left * sb_frac)));
                const int rc = plan_superbatch(qa, sb);
                if (rc == 1) {   // undo what this abandoned attempt counted
                    stats.n_index_entries -= n_entries;
                    stats.algorithmic_bytes[UC_ST_INDEX] -= 6ull * n_entries + 8ull * KSPACE;
                    stats.prefilter_kernel_ms += gpu_ms;
                    return false;
                }
                if (rc == 2) {   // distinct k-mers grow sublinearly with the queries: cut a little deeper than proportionally
                    sb_frac = (double)(sb - qa) / left * 0.8 * (double)DRUN_MAX / (double)over_runs;
                    continue;
                }
                sb_begin = qa; sb_end = sb;
                if (sb_frac < 1.0) sb_frac = std::min(1.0, (double)(sb - qa) / std::max<uint32_t>(1, qend - sb));   // the same number of queries again
                break;
            }
        }
        Timer t_b;
        timed_ms_begin();
        uint32_t qb = qa;
        uint64_t total_hits = 0, n_runs = 0, mirror_hits = 0;
        uint32_t qp0 = 0, qp1 = 0, nq_res = 0;
        {
            // exact plan: as many queries as fit the key and run buffers (a single query may exceed them and takes the wide path)
            while (qb < sb_end && qb - qa < (1u << 23) - 1) {
                const uint64_t h = h_qh[qb - sb_begin], r = h_qr[qb - sb_begin];
                if (qb > qa && (total_hits + h > hit_cap || n_runs + r > RUN_MAX)) break;
                total_hits += h; n_runs += r;
                if (qb >= mirror_q0) mirror_hits += h;       // these hits stand for the hits of the pairs the other way round as well
                qb++;
            }
            if (n_runs >= (1ull << 32)) fail(UC_ERR_GENERIC, "query %u alone produces %llu index ranges", qa, (unsigned long long)n_runs);
            qp0 = h_poff[qa]; qp1 = h_poff[qb]; nq_res = qp1 - qp0;
            run_cap = std::max<uint64_t>(run_cap, n_runs);
            d_rpidx2.reserve(run_cap); d_rval2.reserve(run_cap);
            UC_HIP(hipMemsetAsync(d_counters.p + 3, 0, 32, stream));   // run cursor, batch hits, key cursor, candidate cursor
            if (n_runs)
                hipLaunchKernelGGL(position_expand_kernel, grid_for(nq_res), dim3(256), 0, stream, qp0 - P0, qp1 - P0, S.d_nr.p, S.d_src.p, S.d_cumr.p, S.d_drv2.p,
                                   d_rpidx2.p, d_rval2.p);
        }
        if (total_hits > (1ull << 34)) fail(UC_ERR_GENERIC, "query %u alone produces %llu k-mer hits", qa, (unsigned long long)total_hits);
        n_hits_total += total_hits + mirror_hits;
        uint64_t n_cand = 0;
        if (total_hits) {
            // pass 2: expand runs into keys (filtered to double hits when the rule allows), sort them
            unsigned qbits = 1;
            while ((1u << qbits) < qb - qa) qbits++;
            const unsigned kbits = (unsigned)(fmt.dbits + fmt.tbits) + qbits;
            const uint64_t *sorted = nullptr;
            uint64_t n_sort = 0;
            size_t tb = 0;
            if (p.min_diag_hits >= 2 && total_hits < (1ull << 32)) {
                const uint32_t nq = qb - qa;
                unsigned pbits = 1;
                while ((1ull << pbits) < nq_res) pbits++;
                d_rpidx2.reserve(run_cap); d_rval2.reserve(run_cap);
                d_qbase.reserve(nq); d_qsurv.reserve(nq); d_soff.reserve((size_t)nq + 1);
                // the query regions hold (target, diagonal) keys: u32 in compact mode (half of the u64 buffer stays unused),
                // every region rounded up to KPT keys
                const uint64_t region_cap = total_hits + (uint64_t)KPT * nq;
                d_keys.reserve(fmt.compact ? region_cap : 2 * region_cap);       // regions + survivor area of the same size
                S.d_qr.reserve((size_t)nq + 1);
                hipLaunchKernelGGL(run_range_kernel, grid_for((uint64_t)nq + 1), dim3(256), 0, stream, ddb, qa, nq, qp0, d_rpidx2.p, n_runs, S.d_qr.p);
                S.d_okey.reserve(nq); S.d_okey2.reserve(nq); S.d_oidx.reserve(nq); S.d_order.reserve(nq);
                hipLaunchKernelGGL(run_order_key_kernel, grid_for(nq), dim3(256), 0, stream, nq, S.d_qr.p, S.d_okey.p, S.d_oidx.p);
                UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, S.d_okey.p, S.d_okey2.p, S.d_oidx.p, S.d_order.p, (size_t)nq, 0u, 32u, stream));
                temp_reserve(tb);
                UC_HIP(rocprim::radix_sort_pairs(d_temp.p, tb, S.d_okey.p, S.d_okey2.p, S.d_oidx.p, S.d_order.p, (size_t)nq, 0u, 32u, stream));
                {
                    auto launch = [&](auto kern) {
                        static PerDeviceOnce once[2];
                        once[fmt.compact ? 1 : 0]([&] { UC_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)filter_lds(FILTER_RPT))); });
                        hipLaunchKernelGGL(kern, dim3(nq), dim3(FT), filter_lds(FILTER_RPT), stream, ddb, qa, qp0, d_rpidx2.p, d_rval2.p, S.d_qr.p, S.d_order.p, ent_p, fmt,
                                           d_counters.p + 5, (void *)d_keys.p, region_cap, d_qbase.p, d_qsurv.p);
                    };
                    if (fmt.compact) launch(filter_kernel<true>);
                    else launch(filter_kernel<false>);
                }
                auto sin = rocprim::make_transform_iterator(d_qsurv.p, WidenU32());
                UC_HIP(rocprim::exclusive_scan(nullptr, tb, sin, d_soff.p, (uint64_t)0, (size_t)nq, rocprim::plus<uint64_t>(), stream));
                temp_reserve(tb);
                UC_HIP(rocprim::exclusive_scan(d_temp.p, tb, sin, d_soff.p, (uint64_t)0, (size_t)nq, rocprim::plus<uint64_t>(), stream));
                uint64_t lo = 0; uint32_t ls = 0;
                UC_HIP(hipMemcpyAsync(&lo, d_soff.p + (nq - 1), 8, hipMemcpyDeviceToHost, stream));
                UC_HIP(hipMemcpyAsync(&ls, d_qsurv.p + (nq - 1), 4, hipMemcpyDeviceToHost, stream));
                UC_HIP(hipStreamSynchronize(stream));
                n_sort = lo + ls;
                if (n_sort) {
                    d_keys2.reserve(2 * n_sort);              // dense keys + the sort's output
                    if (fmt.compact)
                        hipLaunchKernelGGL(compact_kernel<true>, dim3(std::min<uint32_t>(nq, 65535u)), dim3(256), 0, stream, (const void *)d_keys.p, d_qbase.p,
                                           d_qsurv.p, d_soff.p, nq, fmt, d_keys2.p);
                    else
                        hipLaunchKernelGGL(compact_kernel<false>, dim3(std::min<uint32_t>(nq, 65535u)), dim3(256), 0, stream, (const void *)d_keys.p, d_qbase.p,
                                           d_qsurv.p, d_soff.p, nq, fmt, d_keys2.p);
                    UC_HIP(rocprim::radix_sort_keys(nullptr, tb, d_keys2.p, d_keys2.p + n_sort, (size_t)n_sort, 0u, kbits, stream));
                    temp_reserve(tb);
                    UC_HIP(rocprim::radix_sort_keys(d_temp.p, tb, d_keys2.p, d_keys2.p + n_sort, (size_t)n_sort, 0u, kbits, stream));
                }
                sorted = d_keys2.p + n_sort;
                stats.n_filtered_hits += n_sort;
            } else {
                d_keys.reserve(total_hits);
                d_keys2.reserve(total_hits);
                const RunList rl{d_rpidx2.p, d_rval2.p, run_cap};
                hipLaunchKernelGGL(expand_kernel, grid_for(n_runs), dim3(256), 0, stream, ddb, qa, qb, qp0, rl, n_runs, ent_p, fmt,
                                   d_counters.p + 5, d_keys.p, total_hits);
                UC_HIP(rocprim::radix_sort_keys(nullptr, tb, d_keys.p, d_keys2.p, (size_t)total_hits, 0u, kbits, stream));
                temp_reserve(tb);
                UC_HIP(rocprim::radix_sort_keys(d_temp.p, tb, d_keys.p, d_keys2.p, (size_t)total_hits, 0u, kbits, stream));
                sorted = d_keys2.p;
                n_sort = total_hits;
                stats.n_filtered_hits += n_sort;
            }
            cand_cap = std::max<uint64_t>(cand_cap, std::max<uint64_t>(1u << 20, total_hits / 32));
            for (;;) {
                d_cq.reserve(cand_cap); d_ct.reserve(cand_cap); d_cd.reserve(cand_cap);
                UC_HIP(hipMemsetAsync(d_counters.p + 6, 0, 8, stream));
                if (n_sort) {
                    S.d_wflag.reserve((n_sort + 63) / 64 + 64);
                    hipLaunchKernelGGL(diag_select_kernel, grid_for(n_sort), dim3(256), 0, stream, sorted, n_sort, p.min_diag_hits, fmt, qa,
                                       d_counters.p + 6, cand_cap, d_cq.p, d_ct.p, d_cd.p, mirror_q0, S.d_wflag.p);
                    hipLaunchKernelGGL(diag_long_kernel, grid_for((n_sort + 63) / 64), dim3(256), 0, stream, sorted, n_sort, p.min_diag_hits, fmt, qa,
                                       d_counters.p + 6, cand_cap, d_cq.p, d_ct.p, d_cd.p, mirror_q0, (const uint8_t *)S.d_wflag.p);
                }
                unsigned long long nc = 0;
                UC_HIP(hipMemcpyAsync(&nc, d_counters.p + 6, 8, hipMemcpyDeviceToHost, stream));
                UC_HIP(hipStreamSynchronize(stream));
                n_cand = nc;
                if (n_cand <= cand_cap) break;
                cand_cap = n_cand;     // rare: more candidates than provisioned, run the selection again
            }
        }
        UC_HIP(hipGetLastError());
        gpu_ms += timed_ms_end();
        t_kmer += t_b.seconds();
        n_cand_total += n_cand;

        if (n_cand) {
            // -------------------------------------------------------- E3: ungapped rescoring
            Timer t_u;
            timed_ms_begin();
            d_score.reserve(n_cand);
            launch_ungapped(ddb, n_cand, d_cq.p, d_ct.p, d_cd.p, d_score.p, d_counters.p + 2, stream);
            UC_HIP(hipGetLastError());
            gpu_ms += timed_ms_end();
            t_ung += t_u.seconds();
            // -------------------------------------------------------- E4: select
            Timer t_s;
            timed_ms_begin();
            d_skey.reserve(n_cand); d_skey2.reserve(n_cand); d_cd2.reserve(n_cand);
            UC_HIP(hipMemsetAsync(d_counters.p + 1, 0, 8, stream));
            hipLaunchKernelGGL(select_key_kernel, grid_for(n_cand), dim3(256), 0, stream, n_cand, d_cq.p, d_ct.p, d_score.p, p.min_ungapped, d_skey.p, d_counters.p + 1);
            size_t tb = 0;
            UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, d_skey.p, d_skey2.p, d_cd.p, d_cd2.p, (size_t)n_cand, 0u, 64u, stream));
            temp_reserve(tb);
            UC_HIP(rocprim::radix_sort_pairs(d_temp.p, tb, d_skey.p, d_skey2.p, d_cd.p, d_cd2.p, (size_t)n_cand, 0u, 64u, stream));
            unsigned long long kept = 0;
            UC_HIP(hipMemcpyAsync(&kept, d_counters.p + 1, 8, hipMemcpyDeviceToHost, stream));
            UC_HIP(hipStreamSynchronize(stream));
            if (kept) {   // rank inside each query's run, keep the first max_seqs, append to the device hit lists
                d_flag.reserve(kept); d_pos.reserve(kept);
                hipLaunchKernelGGL(rank_flag_kernel, grid_for(kept), dim3(256), 0, stream, d_skey2.p, (uint64_t)kept, (uint32_t)p.max_seqs, d_flag.p);
                auto rin = rocprim::make_transform_iterator(d_flag.p, WidenU32());
                UC_HIP(rocprim::exclusive_scan(nullptr, tb, rin, d_pos.p, (uint64_t)0, (size_t)kept, rocprim::plus<uint64_t>(), stream));
                temp_reserve(tb);
                UC_HIP(rocprim::exclusive_scan(d_temp.p, tb, rin, d_pos.p, (uint64_t)0, (size_t)kept, rocprim::plus<uint64_t>(), stream));
                uint64_t lp = 0; uint32_t lf = 0;
                UC_HIP(hipMemcpyAsync(&lp, d_pos.p + (kept - 1), 8, hipMemcpyDeviceToHost, stream));
                UC_HIP(hipMemcpyAsync(&lf, d_flag.p + (kept - 1), 4, hipMemcpyDeviceToHost, stream));
                UC_HIP(hipStreamSynchronize(stream));
                const uint64_t add = lp + lf;
                d_hq.grow_preserve(n_hits + add, n_hits, stream); d_ht.grow_preserve(n_hits + add, n_hits, stream);
                d_hs.grow_preserve(n_hits + add, n_hits, stream); d_hd.grow_preserve(n_hits + add, n_hits, stream);
                hipLaunchKernelGGL(hit_scatter_kernel, grid_for(kept), dim3(256), 0, stream, d_skey2.p, d_cd2.p, (uint64_t)kept, d_flag.p, d_pos.p,
                                   d_hq.p + n_hits, d_ht.p + n_hits, d_hs.p + n_hits, d_hd.p + n_hits);
                n_hits += add;
            }
            UC_HIP(hipGetLastError());
            gpu_ms += timed_ms_end();
            t_sel += t_s.seconds();
        }
        qa = qb;
    }
    // per-query counts / offsets (the host only keeps these two small arrays)
    if (n_hits && !mirrored) {
        d_cnt.reserve(n);
        hipLaunchKernelGGL(hit_count_kernel, grid_for(n), dim3(256), 0, stream, d_hq.p, n_hits, n, d_cnt.p);
        UC_HIP(hipMemcpyAsync(hit_cnt.data(), d_cnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipStreamSynchronize(stream));
    }
    if (!mirrored) {
        for (uint32_t q = 0; q < n; q++) hit_off[q + 1] = hit_off[q] + hit_cnt[q];
        if (hit_off[n] != n_hits) fail(UC_ERR_GENERIC, "prefilter: hit list bookkeeping mismatch");
    }
    unsigned long long ovl = 0;
    UC_HIP(hipMemcpy(&ovl, d_counters.p + 2, 8, hipMemcpyDeviceToHost));
    const uint64_t ungapped_bytes = ovl + 16ull * n_cand_total;   // (overlap + 16) B per candidate, SURVEY.md 8(d)
    stats.n_kmer_hits += n_hits_total;
    stats.n_candidates += n_cand_total;
    stats.algorithmic_bytes[UC_ST_KMER] += 8ull * (stats.n_sim_kmers - sims_before) + 6ull * n_hits_total + 8ull * n_cand_total;
    stats.algorithmic_bytes[UC_ST_UNGAPPED] += ungapped_bytes;
    stats.algorithmic_bytes[UC_ST_SELECT] += 16ull * n_cand_total;
    stats.stage_seconds[UC_ST_KMER] += t_kmer;
    stats.stage_seconds[UC_ST_UNGAPPED] += t_ung;
    stats.stage_seconds[UC_ST_SELECT] += t_sel;
    stats.prefilter_kernel_ms += gpu_ms;
    return true;
}

void Engine::export_hits_dev(uint32_t *dq, uint32_t *dt, int32_t *ds, int32_t *dd) const {
    if (!n_hits) return;
    UC_HIP(hipSetDevice(device));
    UC_HIP(hipMemcpyAsync(dq, d_hq.p, n_hits * 4, hipMemcpyDeviceToDevice, stream));
    UC_HIP(hipMemcpyAsync(dt, d_ht.p, n_hits * 4, hipMemcpyDeviceToDevice, stream));
    UC_HIP(hipMemcpyAsync(ds, d_hs.p, n_hits * 4, hipMemcpyDeviceToDevice, stream));
    UC_HIP(hipMemcpyAsync(dd, d_hd.p, n_hits * 4, hipMemcpyDeviceToDevice, stream));
    UC_HIP(hipStreamSynchronize(stream));
}

// Multi-GPU phase 2: the engine's (merged, truncated) lists regrouped by the rank that owns each pair.  The partition is a
// STABLE radix sort on ceil(log2 world) bits, so every owner's segment keeps the list order (query, score desc, target asc).
// dq..dd: caller-owned device arrays of n_hits elements; counts[world] on the host.
void Engine::partition_hits_by_owner(uint32_t world, uint32_t *dq, uint32_t *dt, int32_t *ds, int32_t *dd, uint64_t *counts) {
    PressureScope ps(*this, 0);
    for (uint32_t w = 0; w < world; w++) counts[w] = 0;
    if (!n_hits) return;
    if (n_hits >= (1ull << 32)) fail(UC_ERR_GENERIC, "partition_hits_by_owner: %llu records exceed the 32-bit index", (unsigned long long)n_hits);
    UC_HIP(hipSetDevice(device));
    if (!pre) pre = take_prefilter_scratch(device);
    DevBuf<uint32_t> &okey = pre->d_cq, &okey2 = pre->d_ct, &idx = pre->d_flag, &idx2 = pre->d_cnt;
    DevBuf<unsigned long long> cnt;
    DevBuf<char> &tmp = pre->d_temp;
    okey.reserve(n_hits); okey2.reserve(n_hits); idx.reserve(n_hits); idx2.reserve(n_hits); cnt.reserve(world);
    UC_HIP(hipMemsetAsync(cnt.p, 0, (size_t)world * 8, stream));
    hipLaunchKernelGGL(owner_key_kernel, grid_for(n_hits), dim3(256), 0, stream, n_hits, d_hq.p, d_ht.p, ddb.len, world, okey.p, idx.p, cnt.p);
    unsigned bits = 1;
    while ((1u << bits) < world) bits++;
    size_t tb = 0;
    UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, okey.p, okey2.p, idx.p, idx2.p, (size_t)n_hits, 0u, bits, stream));
    tmp.reserve(tb + 256);
    UC_HIP(rocprim::radix_sort_pairs(tmp.p, tb, okey.p, okey2.p, idx.p, idx2.p, (size_t)n_hits, 0u, bits, stream));
    hipLaunchKernelGGL(owner_gather_kernel, grid_for(n_hits), dim3(256), 0, stream, n_hits, idx2.p, d_hq.p, d_ht.p, d_hs.p, d_hd.p, dq, dt, ds, dd);
    std::vector<unsigned long long> hc(world);
    UC_HIP(hipMemcpyAsync(hc.data(), cnt.p, (size_t)world * 8, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipStreamSynchronize(stream));
    UC_HIP(hipGetLastError());
    for (uint32_t w = 0; w < world; w++) counts[w] = hc[w];
}

uint64_t Engine::import_hits_dev(uint64_t n_in, const uint32_t *dq, const uint32_t *dt, const int32_t *ds, const int32_t *dd,
                                 uint32_t rank, uint32_t world) {
    return merge_hits_dev(0, nullptr, nullptr, nullptr, nullptr, n_in, dq, dt, ds, dd, false, rank, world);
}

// The union of two record lists, merged per query under the frozen order (score desc, target asc), truncated to max_seqs and installed.  List 1 is
// ALREADY in key order (a running top-M accumulator, i.e. the output of an earlier merge or a grouped pass); list 2 is sorted here unless the caller says
// it is in key order as well.  r06: the chunk loop of prefilter_impl used to radix-sort accumulator + chunk after every pass - at nominal configs[3] the
// accumulator of up to max_seqs x queries records went through ~7 sort passes (~170 B of traffic per record) once per target chunk, 8.9 s of the call.
// Now only the chunk's records are sorted and the two runs are merged (one read + one write of every record); the result is the stable sort's, record for record
// (the passes' candidate sets are disjoint, so no two keys are equal; rocprim::merge takes list 1 first on ties, as the stable sort of the concatenation did).
uint64_t Engine::merge_hits_dev(uint64_t n1, const uint32_t *q1, const uint32_t *t1, const int32_t *s1, const int32_t *d1,
                                uint64_t n2, const uint32_t *q2, const uint32_t *t2, const int32_t *s2, const int32_t *d2, bool sorted2,
                                uint32_t rank, uint32_t world) {
    PressureScope ps(*this, 0);
    if (!have_db) fail(UC_ERR_ARGS, "no database loaded");
    UC_HIP(hipSetDevice(device));
    const uint32_t n = hdb.n;
    if (n > (1u << 24)) fail(UC_ERR_GENERIC, "hits_import_dev: %u sequences exceed the 2^24 limit of the hit keys", n);
    hit_cnt.assign(n, 0);
    hit_off.assign((size_t)n + 1, 0);
    n_hits = 0;
    alns_valid = false;
    clear_edges();
    const uint64_t n_in = n1 + n2;
    if (!n_in) return 0;
    Timer tm;
    timed_ms_begin();
    // work buffers from the engine's prefilter scratch: a chunked or sharded run merges once per chunk / shard, and allocating
    // ~50 B per record anew every time was most of the merge time at 10^9 records
    if (!pre) pre = take_prefilter_scratch(device);
    DevBuf<uint64_t> &skey = pre->d_skey, &skey2 = pre->d_skey2, &pos = pre->d_pos;
    DevBuf<int32_t> &cd2 = pre->d_cd2;
    DevBuf<uint32_t> &flag = pre->d_flag, &cnt = pre->d_cnt;
    DevBuf<uint32_t> bad;
    DevBuf<char> &tmp = pre->d_temp;
    // the merge path (rocprim::merge partitions with 32-bit indices) - UC_MERGE_SORTED=0 keeps the plain sort of everything for A/B runs
    static const bool merge_ok = !(getenv("UC_MERGE_SORTED") && atoi(getenv("UC_MERGE_SORTED")) == 0);
    const bool merging = n1 > 0 && n2 > 0 && merge_ok && n_in < (1ull << 32) - (1ull << 20);
    const bool presorted = n1 > 0 && n2 == 0 && merge_ok;              // nothing to add: only truncation / ownership / counts
    skey.reserve(n_in); flag.reserve(n_in); pos.reserve(n_in); bad.reserve(1);
    UC_HIP(hipMemsetAsync(bad.p, 0, 4, stream));
    if (n1) hipLaunchKernelGGL(merge_key_kernel, grid_for(n1), dim3(256), 0, stream, n1, q1, t1, s1, n, skey.p, bad.p);
    if (n2) hipLaunchKernelGGL(merge_key_kernel, grid_for(n2), dim3(256), 0, stream, n2, q2, t2, s2, n, skey.p + n1, bad.p);
    size_t tb = 0;
    unsigned kb = 33;                              // [query | 255 - score : 8 | target : 24]: the query field ends at bit 32 + log2 n
    while (kb < 64 && (1ull << (kb - 32)) < n) kb++;
    const uint64_t *sk = nullptr;                  // the sorted keys and their diagonals
    const int32_t *sd = nullptr;
    if (presorted) { sk = skey.p; sd = d1; }
    else if (merging) {
        const uint64_t *k2 = skey.p + n1;
        const int32_t *v2 = d2;
        if (!sorted2) {
            skey2.reserve(n2); cd2.reserve(n2);
            UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, skey.p + n1, skey2.p, d2, cd2.p, (size_t)n2, 0u, kb, stream));
            tmp.reserve(tb + 256);
            UC_HIP(rocprim::radix_sort_pairs(tmp.p, tb, skey.p + n1, skey2.p, d2, cd2.p, (size_t)n2, 0u, kb, stream));
            k2 = skey2.p; v2 = cd2.p;
        }
        DevBuf<uint64_t> &mkey = pre->d_mkey;
        DevBuf<int32_t> &mval = pre->d_mval;
        mkey.reserve(n_in); mval.reserve(n_in);
        UC_HIP(rocprim::merge(nullptr, tb, skey.p, k2, mkey.p, d1, v2, mval.p, (size_t)n1, (size_t)n2, rocprim::less<uint64_t>(), stream));
        tmp.reserve(tb + 256);
        UC_HIP(rocprim::merge(tmp.p, tb, skey.p, k2, mkey.p, d1, v2, mval.p, (size_t)n1, (size_t)n2, rocprim::less<uint64_t>(), stream));
        sk = mkey.p; sd = mval.p;
    } else {
        // one list (or the A/B switch): the diagonals of the two inputs have to be one array for the pair sort
        const int32_t *dd = n1 ? nullptr : d2;
        skey2.reserve(n_in); cd2.reserve(n_in);
        if (n1) {
            DevBuf<int32_t> &mval = pre->d_mval;
            mval.reserve(n_in);
            UC_HIP(hipMemcpyAsync(mval.p, d1, n1 * 4, hipMemcpyDeviceToDevice, stream));
            if (n2) UC_HIP(hipMemcpyAsync(mval.p + n1, d2, n2 * 4, hipMemcpyDeviceToDevice, stream));
            dd = mval.p;
        }
        UC_HIP(rocprim::radix_sort_pairs(nullptr, tb, skey.p, skey2.p, dd, cd2.p, (size_t)n_in, 0u, kb, stream));
        tmp.reserve(tb + 256);
        UC_HIP(rocprim::radix_sort_pairs(tmp.p, tb, skey.p, skey2.p, dd, cd2.p, (size_t)n_in, 0u, kb, stream));
        sk = skey2.p; sd = cd2.p;
    }
    hipLaunchKernelGGL(rank_flag_kernel, grid_for(n_in), dim3(256), 0, stream, sk, n_in, (uint32_t)p.max_seqs, flag.p);
    if (world > 1) hipLaunchKernelGGL(owner_flag_kernel, grid_for(n_in), dim3(256), 0, stream, sk, n_in, ddb.len, rank, world, flag.p);
    auto rin = rocprim::make_transform_iterator(flag.p, WidenU32());
    UC_HIP(rocprim::exclusive_scan(nullptr, tb, rin, pos.p, (uint64_t)0, (size_t)n_in, rocprim::plus<uint64_t>(), stream));
    tmp.reserve(tb + 256);
    UC_HIP(rocprim::exclusive_scan(tmp.p, tb, rin, pos.p, (uint64_t)0, (size_t)n_in, rocprim::plus<uint64_t>(), stream));
    uint64_t lp = 0; uint32_t lf = 0, hbad = 0;
    UC_HIP(hipMemcpyAsync(&lp, pos.p + (n_in - 1), 8, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipMemcpyAsync(&lf, flag.p + (n_in - 1), 4, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipMemcpyAsync(&hbad, bad.p, 4, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipStreamSynchronize(stream));
    if (hbad) fail(UC_ERR_ARGS, "hits_import_dev: %u records with a sequence id or score out of range", hbad);
    const uint64_t keep = lp + lf;
    if (keep) {
        // (list 1 or 2 may BE the engine's own arrays: the scatter reads only keys and the sorted diagonals, never q / t / s / d of the inputs - but the
        // diagonals of a presorted or merged input are read from d1 / d2 themselves, so the output must not overwrite them in place)
        const bool alias = (sd == d1 && d1 == d_hd.p) || (sd == d2 && d2 == d_hd.p);
        if (alias) {
            DevBuf<int32_t> &mval = pre->d_mval;
            mval.reserve(n_in);
            UC_HIP(hipMemcpyAsync(mval.p, sd, n_in * 4, hipMemcpyDeviceToDevice, stream));
            sd = mval.p;
        }
        d_hq.reserve(keep); d_ht.reserve(keep); d_hs.reserve(keep); d_hd.reserve(keep);
        hipLaunchKernelGGL(hit_scatter_kernel, grid_for(n_in), dim3(256), 0, stream, sk, sd, n_in, flag.p, pos.p, d_hq.p, d_ht.p, d_hs.p, d_hd.p);
        n_hits = keep;
        cnt.reserve(n);
        hipLaunchKernelGGL(hit_count_kernel, grid_for(n), dim3(256), 0, stream, d_hq.p, n_hits, n, cnt.p);
        UC_HIP(hipMemcpyAsync(hit_cnt.data(), cnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipStreamSynchronize(stream));
    }
    UC_HIP(hipGetLastError());
    for (uint32_t q = 0; q < n; q++) hit_off[q + 1] = hit_off[q] + hit_cnt[q];
    if (hit_off[n] != n_hits) fail(UC_ERR_GENERIC, "hits_import_dev: hit list bookkeeping mismatch");
    stats.prefilter_kernel_ms += timed_ms_end();
    stats.stage_seconds[UC_ST_SELECT] += tm.seconds();
    return keep;
}

void preload_prefilter_module() { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)kmer_extract_kernel); }
}  // namespace uc
