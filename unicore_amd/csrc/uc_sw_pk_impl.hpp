// uc_sw_pk_impl.hpp — packed 16-bit variant of the gapped DP kernel (stage E5; see uc_sw_impl.hpp for the
// systolic-group design it shares).  The int32 kernel is integer-VALU-bound (every int32 VALU instruction
// occupies its SIMD for 4 cycles, profiles/round1/r1b_pmc_sw.txt), so the only lever left is instructions per cell:
// here every DP register holds TWO alignments of the same query (target A in the low, target B in the high
// 16 bits) and the recurrence runs on v_pk_{add,sub,max}_u16, i.e. ~6.3 VALU ops per cell instead of ~9.7.
//
//  * unsigned floored domain: H >= 0, E/F/T floored at 0 by saturating subtraction (exact for local
//    alignment); x = sat_sub(sat_add(Hdiag, s + 128), 128) = max(Hdiag + s, 0); H = max3(x, e, f) in one
//    v_pk_maximum3_f16 (see SW_PK_OVF).
//  * profile bytes are (S3+64) and (SA+64) so the packed byte sum is s+128 (PAD = 0 => s = -128); one
//    v_perm_b32 per row interleaves byte r of target A's and target B's word into two zero-extended u16.
//  * end tracking: per lane the running maximum gives (score, first column) per half; per-row maxima
//    (rowbest) identify the row: if exactly one row of the alignment ever reaches the optimum it must be the
//    row of the first column too, so (qEnd, tEnd) is exact.  Otherwise — or if a value came within 256 of the
//    packed score range (0x7C00, see SW_PK_OVF) — the pair is flagged (qEnd = -2 / score >= SW_PK_OVF): overflows are
//    re-run by the int32 kernel at once, ambiguous end rows only if the pair passes the E-value gate, by the
//    known-score variant of this kernel (MODE 4 / 6).
//  * slot streaming: a "slot" is two consecutive pairs of the task (A, B).  Every lane group pulls its next
//    slot from an LDS counter as soon as it has finished the previous one, so the groups of a wave do NOT
//    run in lockstep over the longest of their targets (hit lists mix family members with unrelated hits of
//    very different lengths; lockstep cost ~40 % of the issued instructions, profiles/round1/r1e).
#pragma once
#include <hip/hip_runtime.h>

#include "uc_device.h"
#include "uc_sw_impl.hpp"

namespace uc {

// H = max(x, e, f) is ONE instruction: v_pk_maximum3_f16 orders non-negative 16-bit integers below 0x7C00 (the f16
// infinity pattern) exactly like an unsigned max, denormal range included, at full rate, and returns >= 0x7C00 as
// soon as an operand is >= 0x7C00 (tools/ubench/pk_max3_f16.hip: exhaustive check on gfx950) - so the packed score
// range ends at 0x7C00 and an overflow is never lost (best/colmax are integer maxima, the flag is sticky).
constexpr int SW_PK_OVF = 0x7C00 - 256;   // scores at or above this are recomputed in int32

typedef uint16_t u16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned wide stores (MODE 7)
typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef const uint32_t __attribute__((address_space(4))) *sw_cu32p;         // constant address space: uniform loads become s_load_dword
__device__ __forceinline__ u16x2_t pk_v(uint32_t x) { return __builtin_bit_cast(u16x2_t, x); }
__device__ __forceinline__ uint32_t pk_u(u16x2_t v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b) { return pk_u(__builtin_elementwise_add_sat(pk_v(a), pk_v(b))); }
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) { return pk_u(__builtin_elementwise_sub_sat(pk_v(a), pk_v(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return pk_u(__builtin_elementwise_max(pk_v(a), pk_v(b))); }
// inline asm: the compiler canonicalises min(x, 1) into compare + select per half (13 instructions instead of 1)
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int G, int R, int MODE, int NW>
__global__ void __launch_bounds__(NW * 64) sw_pk_kernel(const SwArgs a) {
    // MODE 4 / 6 = MODE 0 / 2 with the optimum score of every pair KNOWN (a.pscore).  MODE 4 is the exact re-run of the
    // pairs whose end row was ambiguous; MODE 6 is THE start pass of the packed path (its optimum is the forward score).
    // No per-row maxima: the first step at which a lane's column maximum equals the known score is its first optimal
    // column, and only at such steps (a rare, wave-level branch) the lane looks at its rows: first optimal row, and
    // whether a single row holds all optimal cells (the condition under which a mutual hit may share the result).
    // MODE 7 = traceback bytes: the forward DP on the box [qs..qe] x [ts..te] of an accepted pair; instead of tracking an
    // end position every cell of the diagonal band the traceback can reach (tb_band_of, uc_device.h; r05) stores ONE byte, H mod 256,
    // into a per-pair matrix in HBM, laid out by anti-diagonal step so that the lanes inside the band store NL*RB contiguous bytes per step.  No decision is computed here (r04: six compare bits per
    // cell had cost 16 of the pass's 27 VALU instructions per row and step; a byte of H costs the byte shuffle only).  A
    // second kernel (tb_walk_kernel, uc_align.hip) walks every pair's matrix from the end cell, whose H it knows (the
    // score): neighbouring cells differ by less than 128, so every H it needs is exact again, and every decision of the
    // traceback (diagonal / F / E, gap opened or extended) is a comparison of H values: alignment length, identities, gaps.
    constexpr bool TBB = MODE == 7;
    constexpr bool KNOWN = MODE == 4 || MODE == 6;
    constexpr int BASE = KNOWN ? MODE - 4 : (TBB ? 0 : MODE);
    constexpr bool TRACK = BASE != 1 && !TBB, MASK = BASE == 2 || TBB, REVQ = BASE != 0, REVT = BASE == 2;
    constexpr int RW = (R + 3) / 4, BW = RW | 1, RSW = G * BW, NT = NW * 64;   // R % 4 == 2: the last profile dword is half used
    static_assert(R % 2 == 0 && R <= 32, "R must be even, <= 32");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *P3 = lds, *PA = lds + SW_NLET * RSW;
    uint32_t *slot_ctr = lds + 2 * SW_NLET * RSW;   // work counter of the task (one dword behind the profile)

    const SwTask task = a.tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = lane % G;
    const uint32_t qoff = a.db.off[task.q];
    const int lq = (int)a.db.len[task.q];
    const uint32_t open2 = (uint32_t)a.open * 0x10001u, ext2 = (uint32_t)a.ext * 0x10001u, bias2 = 128u * 0x10001u;
    const uint32_t nslots = (task.count + 1) >> 1;

    // ---- query profile in LDS: bytes S3+64 and SA+64 (sum = s+128), PAD letters / rows = 0 ----
    // (r06, measured and not kept: the two 21 x 21 matrices and the query letters staged in LDS first, the entries assembled from LDS bytes instead of eight
    // dependent global round trips each - no change in any pass, the sparse ones included (mode 4 at configs[1]: 23.1 ms either way): the build is not
    // what a task waits for.  profiles/r06/sparse_passes.txt)
    for (int idx = tid; idx < SW_NLET * G * RW; idx += NT) {
        const int c = idx / (G * RW), rem = idx % (G * RW), gg = rem / RW, k = rem % RW;
        uint32_t w3 = 0, wa = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int row = gg * R + 4 * k + b;
            if (row < lq && c < 21 && 4 * k + b < R) {
                const int qi = REVQ ? lq - 1 - row : row;
                const int q3 = a.db.s3[qoff + qi], qa = a.db.sa[qoff + qi];
                w3 |= (uint32_t)(a.db.S3[q3 * 21 + c] + 64) << (8 * b);
                wa |= (uint32_t)(a.db.SA[qa * 21 + c] + 64) << (8 * b);
            }
        }
        P3[c * RSW + gg * BW + k] = w3;
        PA[c * RSW + gg * BW + k] = wa;
    }
    if (tid == 0) *slot_ctr = 0;
    __syncthreads();

    // ---- per-group state (all group-uniform scalars live replicated in the group's lanes) ----
    uint32_t H[R], E[R], rowbest[(TRACK && !KNOWN) ? R : 1];   // E[r] holds the gap state ENTERING the next column (no separate H - open array)
    uint32_t mskA[MASK ? RW : 1], mskB[MASK ? RW : 1];
    uint32_t best = 0, Hlast = 0, prevHup = 0, fout = 0;
    int colA = -1, colB = -1;
    [[maybe_unused]] int rowA = 0, rowB = 0;
    [[maybe_unused]] bool multA = false, multB = false;   // KNOWN: this lane saw the optimum in more than one of its rows
    [[maybe_unused]] uint32_t knownA = 0, knownB = 0;
    uint32_t gA = 0, gB = 0, toffA = 0, toffB = 0;
    [[maybe_unused]] unsigned long long tbA = 0, tbB = 0;
    [[maybe_unused]] int tsaA = 0, tsbA = 0, tsaB = 0, tsbB = 0;        // MODE 7: the steps in which this lane is inside the pair's stored band
    [[maybe_unused]] uint32_t trowA = 0, trowB = 0, tslotA = 0, tslotB = 0;   // ... bytes per step of the pair's matrix, this lane's byte offset in a step
    int tlenA = 0, tlenB = 0, rowoffA = 0, rowoffB = 0;
    bool vB = false, active = false;
    int lst = 0, nst = 0;
    uint32_t c1 = 0, cin = 0;
    struct RawLetters { uint32_t a3, aa, b3, ba; } r2 = {0, 0, 0, 0};   // G < 64: a3 / b3 hold the whole 16-bit letter pair
    uint32_t nA3[RW], nAa[RW], nB3[RW], nBa[RW];
    // G < 64: both letters of a residue come from the interleaved stream a.db.lt (3Di | AA << 8, PAD pairs between the
    // sequences): one 16-bit load per target and step, the index clamped onto the PAD entry behind (REVT: before) the
    // sequence, so neither a second byte load nor a "past the end" select is needed.  ltA / ltB point at element 0
    // (REVT: at the PAD entry before it, so that the index stays unsigned).
    [[maybe_unused]] const uint16_t *ltA = nullptr, *ltB = nullptr;

    // letters of both targets travel as one dword: [c3A | caA << 8 | c3B << 16 | caB << 24].  The loads are
    // branch-free (clamped index, every lane of the group reads the same address) and are only combined a full
    // step after they were issued, so no step waits on global-memory latency.
    auto issue_letters = [&](int st) -> RawLetters {
        RawLetters r;
        if constexpr (G == 64) {   // one group per wave: the slot state is wave-uniform, so the letters come through the scalar
                                   // cache (aligned dword + shift on the SALU) and cost no VALU issue slots at all
            const int ia = max(min(st, tlenA - 1), 0), ib = max(min(st, tlenB - 1), 0);
            const uint32_t pa = toffA + (uint32_t)(REVT ? max(tlenA - 1 - ia, 0) : ia);
            const uint32_t pb_ = toffB + (uint32_t)(REVT ? max(tlenB - 1 - ib, 0) : ib);
            auto ld = [&](const uint8_t *base, uint32_t p) -> uint32_t {
                const sw_cu32p w = (sw_cu32p)(uintptr_t)(base + (p & ~3u));
                return (*w >> (8u * (p & 3u))) & 0xffu;
            };
            r.a3 = ld(a.db.s3, pa); r.aa = ld(a.db.sa, pa); r.b3 = ld(a.db.s3, pb_); r.ba = ld(a.db.sa, pb_);
        } else {
            const uint32_t ia = REVT ? (uint32_t)max(tlenA - st, 0) : (uint32_t)min(st, tlenA);
            const uint32_t ib = REVT ? (uint32_t)max(tlenB - st, 0) : (uint32_t)min(st, tlenB);
            r.a3 = ltA[ia]; r.b3 = ltB[ib];
            r.aa = 0; r.ba = 0;
        }
        return r;
    };
    auto pack_letters = [&](const RawLetters &r, int st) -> uint32_t {
        if constexpr (G == 64) {
            const uint32_t ca_ = st < tlenA ? (r.a3 | (r.aa << 8)) : SW_PADPACK;
            const uint32_t cb_ = st < tlenB ? (r.b3 | (r.ba << 8)) : SW_PADPACK;
            return ca_ | (cb_ << 16);
        } else if constexpr (TBB) {   // a box ends inside its sequence: the columns past it are PAD by selection
            return (st < tlenA ? r.a3 : SW_PADPACK) | ((st < tlenB ? r.b3 : SW_PADPACK) << 16);
        } else {
            return r.a3 | (r.b3 << 16);
        }
    };
    uint32_t off3 = (uint32_t)(g * BW) * 4u, offa = (uint32_t)(SW_NLET * RSW + g * BW) * 4u;
    asm volatile("" : "+v"(off3), "+v"(offa));
    auto fetch_profile = [&]() __attribute__((always_inline)) {
        // byte offsets: letter * row-set stride + this lane's base in P3 / PA (two opaque registers, so that the PA base is
        // the addend of the multiply-add instead of a separate add per load)
        const char *l8 = (const char *)lds;
        const uint32_t *pA3 = (const uint32_t *)(l8 + (cin & 0xff) * (RSW * 4) + off3), *pAa = (const uint32_t *)(l8 + ((cin >> 8) & 0xff) * (RSW * 4) + offa);
        const uint32_t *pB3 = (const uint32_t *)(l8 + ((cin >> 16) & 0xff) * (RSW * 4) + off3), *pBa = (const uint32_t *)(l8 + (cin >> 24) * (RSW * 4) + offa);
#pragma unroll
        for (int k = 0; k < RW; k++) { nA3[k] = pA3[k]; nAa[k] = pAa[k]; nB3[k] = pB3[k]; nBa[k] = pBa[k]; }
    };

    // pull the next slot of the task for this group and reset the group's DP state (group-uniform control flow)
    auto start_slot = [&]() {
        uint32_t idx = 0;
        if (g == 0) idx = atomicAdd(slot_ctr, 1u);
        if constexpr (G == 64) idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);   // wave-uniform from here on (SGPRs)
        else idx = (uint32_t)__shfl((int)idx, lane - g, 64);
        active = idx < nslots;
        if (!active) return;
        const uint32_t iA = 2 * idx, iB = iA + 1;
        vB = iB < task.count;
        gA = task.begin + iA;
        gB = task.begin + (vB ? iB : iA);
        const uint32_t tA = a.pt[gA], tB = a.pt[gB];
        toffA = a.db.off[tA]; toffB = a.db.off[tB];
        tlenA = REVT ? a.pte[gA] + 1 : (int)a.db.len[tA];
        tlenB = vB ? (REVT ? a.pte[gB] + 1 : (int)a.db.len[tB]) : 0;
        if constexpr (TBB) {   // the box: target slice [ts, te], query rows [qs, qe]
            toffA += (uint32_t)a.pts[gA]; toffB += (uint32_t)a.pts[gB];
            tlenA = a.pte[gA] - a.pts[gA] + 1;
            tlenB = vB ? a.pte[gB] - a.pts[gB] + 1 : 0;
            tbA = a.tboff[gA]; tbB = a.tboff[gB];
            const int qsA = a.pqs[gA], qeA = a.pqe[gA], qsB = a.pqs[gB], qeB = a.pqe[gB];
            {   // the stored band of either pair (tb_band_of, uc_device.h)
                constexpr int RBc = 4 * RW;
                const TbBand bA = tb_band_of(qsA, qeA, tlenA, G, R, a.tb_band), bB = tb_band_of(qsB, qeB, vB ? tlenB : 1, G, R, a.tb_band);
                const bool fullA = bA.nl >= G, fullB = bB.nl >= G;
                tsaA = fullA ? 0 : g * (R + 1) - bA.dhi; tsbA = fullA ? 0x7fffffff : g * (R + 1) + R - 1 - bA.dlo;
                tsaB = fullB ? 0 : g * (R + 1) - bB.dhi; tsbB = fullB ? 0x7fffffff : g * (R + 1) + R - 1 - bB.dlo;
                trowA = (uint32_t)(bA.nl * RBc); trowB = (uint32_t)(bB.nl * RBc);
                tslotA = (uint32_t)((g % bA.nl) * RBc); tslotB = (uint32_t)((g % bB.nl) * RBc);
            }
#pragma unroll
            for (int k = 0; k < RW; k++) {
                uint32_t ma = 0, mb = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int row = g * R + 4 * k + b;
                    ma |= (row >= qsA && row <= qeA ? 0xFFu : 0u) << (8 * b);
                    mb |= (row >= qsB && row <= qeB ? 0xFFu : 0u) << (8 * b);
                }
                mskA[k] = ma; mskB[k] = mb;
            }
        } else if constexpr (MASK) {
            rowoffA = lq - 1 - a.pqe[gA];
            rowoffB = lq - 1 - a.pqe[gB];
#pragma unroll
            for (int k = 0; k < RW; k++) {
                uint32_t ma = 0, mb = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    ma |= (g * R + 4 * k + b >= rowoffA ? 0xFFu : 0u) << (8 * b);
                    mb |= (g * R + 4 * k + b >= rowoffB ? 0xFFu : 0u) << (8 * b);
                }
                mskA[k] = ma; mskB[k] = mb;
            }
        }
        if constexpr (G != 64) {
            ltA = a.db.lt + toffA - (REVT ? 1 : 0);
            // no second pair: every index lands on a PAD entry (the one before A's first residue)
            ltB = vB ? a.db.lt + toffB - (REVT ? 1 : 0) : a.db.lt + toffA - 1;
        }
#pragma unroll
        for (int r = 0; r < R; r++) { H[r] = 0; E[r] = 0; }
        if constexpr (TRACK && !KNOWN) {
#pragma unroll
            for (int r = 0; r < R; r++) rowbest[r] = 0;
        }
        if constexpr (KNOWN) {
            knownA = (uint32_t)a.pscore[gA];
            knownB = vB ? (uint32_t)a.pscore[gB] : 0u;
            rowA = 0; rowB = 0; multA = false; multB = false;
        }
        best = 0; colA = -1; colB = -1; Hlast = 0; prevHup = 0; fout = 0;
        lst = 0;
        nst = (max(tlenA, tlenB) + G) & ~1;      // Lt + G - 1 steps, rounded up to the 2-step loop trip
        c1 = pack_letters(issue_letters(1), 1);
        r2 = issue_letters(2);
        if constexpr (G == 64) {   // keep the letter pipeline in SGPRs across the loop back edge
            c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c1);
            r2.a3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r2.a3); r2.aa = (uint32_t)__builtin_amdgcn_readfirstlane((int)r2.aa);
            r2.b3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r2.b3); r2.ba = (uint32_t)__builtin_amdgcn_readfirstlane((int)r2.ba);
        }
        cin = (uint32_t)shift_from_prev_lane<G>((int)(SW_PADPACK * 0x10001u), (int)pack_letters(issue_letters(0), 0), g);
        fetch_profile();
    };

    // ODD = second step of the 2-step loop trip: the per-row maxima take this step's and the previous step's H in one max3
    auto do_step = [&](const int st, const bool ODD) __attribute__((always_inline)) {
        uint32_t sA[RW], sB[RW];
#pragma unroll
        for (int k = 0; k < RW; k++) {
            sA[k] = nA3[k] + nAa[k];
            sB[k] = nB3[k] + nBa[k];
            if constexpr (MASK) { sA[k] &= mskA[k]; sB[k] &= mskB[k]; }
        }
        cin = (uint32_t)shift_from_prev_lane<G>((int)cin, (int)c1, g);
        c1 = pack_letters(r2, st + 2);
        r2 = issue_letters(st + 3);
        fetch_profile();
        const uint32_t Hup = (uint32_t)shift_from_prev_lane_zero<G>((int)Hlast, g);
        uint32_t f = (uint32_t)shift_from_prev_lane_zero<G>((int)fout, g);
        uint32_t diag = prevHup;
        uint32_t colmax = 0;
        [[maybe_unused]] uint32_t code[TBB ? R : 1];
#pragma unroll
        for (int r = 0; r < R; r++) {
            // {byte r of A's word, 0, byte r of B's word, 0}
            const uint32_t ub = __builtin_amdgcn_perm(sB[r >> 2], sA[r >> 2], 0x0c000c00u | ((4u + (r & 3)) << 16) | (uint32_t)(r & 3));
            const uint32_t x = pk_sub_sat(pk_add_sat(diag, ub), bias2);
            const uint32_t e = E[r];                          // max(E - ext, H - open) of the previous column
            const uint32_t h = pk_max3(x, e, f);
            const uint32_t hprev = H[r];                      // this row one column earlier
            diag = hprev;
            H[r] = h;
            const uint32_t t = pk_sub_sat(h, open2);
            const uint32_t esub = pk_sub_sat(e, ext2), fsub = pk_sub_sat(f, ext2);
            if constexpr (TBB) code[r] = h;   // the cell's byte = H mod 256 of either half (packed into byte streams below)
            E[r] = pk_max(esub, t);
            f = pk_max(fsub, t);
            if constexpr (TRACK && !KNOWN) { if (ODD) rowbest[r] = pk_max3(rowbest[r], hprev, h); }
            if (r & 1) colmax = pk_max3(colmax, H[r - 1], h);   // two rows per instruction (R is even)
        }
        if constexpr (TBB) {   // bytes of 4 rows -> one dword per pair; step-major matrix: G*RB contiguous bytes per group and step
            uint32_t wa[RW], wb[RW];
#pragma unroll
            for (int k = 0; k < RW; k++) {
                const uint32_t c0 = code[4 * k], c1_ = 4 * k + 1 < R ? code[4 * k + 1] : 0u;
                const uint32_t c2 = 4 * k + 2 < R ? code[4 * k + 2] : 0u, c3 = 4 * k + 3 < R ? code[4 * k + 3] : 0u;
                const uint32_t p01 = __builtin_amdgcn_perm(c1_, c0, 0x06020400u);   // [A0, A1, B0, B1]
                const uint32_t p23 = __builtin_amdgcn_perm(c3, c2, 0x06020400u);    // [A2, A3, B2, B3]
                wa[k] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                wb[k] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
            }
            // every lane writes its RB bytes with as few (dword-aligned) wide stores as possible: the lanes of a group cover
            // G*RB contiguous bytes, so the stores of a step fill whole lines; streaming (written once, read sparsely)
            // (a step's row holds the lanes inside the pair's band only: trow = NL * RB bytes, this lane at tslot)
            uint8_t *dA = a.tbm + tbA + (unsigned long long)st * trowA + tslotA;
            uint8_t *dB = a.tbm + tbB + (unsigned long long)st * trowB + tslotB;
            auto put = [&](uint8_t *d, const uint32_t *w) __attribute__((always_inline)) {
                int k = 0;
#pragma unroll
                for (; k + 4 <= RW; k += 4) {
                    u32x4_u v = {w[k], w[k + 1], w[k + 2], w[k + 3]};
                    *(u32x4_u *)(d + 4 * k) = v;
                }
#pragma unroll
                for (; k + 2 <= RW; k += 2) {
                    u32x2_u v = {w[k], w[k + 1]};
                    *(u32x2_u *)(d + 4 * k) = v;
                }
#pragma unroll
                for (; k < RW; k++) *(uint32_t *)(d + 4 * k) = w[k];
            };
            if (st >= tsaA && st <= tsbA) put(dA, wa);
            if (vB && st >= tsaB && st <= tsbB) put(dB, wb);
        }
        if constexpr (TRACK && !KNOWN) {
            const uint32_t cmA = colmax & 0xffffu, cmB = colmax >> 16;
            colA = cmA > (best & 0xffffu) ? st - g : colA;
            colB = cmB > (best >> 16) ? st - g : colB;
        }
        if constexpr (KNOWN) {
            // every column of this lane that holds the optimum is an event (a handful per alignment: wave-level branch);
            // the first one gives (column, row); all of them together tell whether ONE row holds every optimal cell
            const bool evA = knownA != 0 && (colmax & 0xffffu) == knownA;
            const bool evB = knownB != 0 && (colmax >> 16) == knownB;
            if (__builtin_amdgcn_ballot_w64(evA || evB) != 0) {
                if (evA) {
                    int first = 0, cnt = 0;
#pragma unroll
                    for (int r = R - 1; r >= 0; r--) { const bool hit = (H[r] & 0xffffu) == knownA; first = hit ? r : first; cnt += hit ? 1 : 0; }
                    if (colA < 0) { colA = st - g; rowA = first; }
                    multA |= cnt > 1 || first != rowA;
                }
                if (evB) {
                    int first = 0, cnt = 0;
#pragma unroll
                    for (int r = R - 1; r >= 0; r--) { const bool hit = (H[r] >> 16) == knownB; first = hit ? r : first; cnt += hit ? 1 : 0; }
                    if (colB < 0) { colB = st - g; rowB = first; }
                    multB |= cnt > 1 || first != rowB;
                }
            }
        }
        best = pk_max(best, colmax);
        Hlast = H[R - 1];
        fout = f;
        prevHup = Hup;
    };

    // per-half reduction over the G lanes: (score desc, col asc); then the row from rowbest; write results
    auto finish_slot = [&]() {
        if constexpr (TBB) return;
        if constexpr (KNOWN) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int col = half ? colB : colA, row = half ? rowB : rowA;
                int key = col < 0 ? 0x7fffffff : ((col << 11) | (g * R + row));      // (first optimal column, then first row)
                // lanes that saw the optimum, + G if one of them saw it in two rows: exactly 1 <=> one row holds every optimal cell
                int nrows = (col < 0 ? 0 : 1) + ((half ? multB : multA) ? G : 0);
#pragma unroll
                for (int m = 1; m < G; m <<= 1) {
                    key = min(key, __shfl_xor(key, m, 64));
                    nrows += __shfl_xor(nrows, m, 64);
                }
                const bool valid = half ? vB : true;
                const uint32_t gp = half ? gB : gA;
                if (g == 0 && valid) {
                    const int rowoff = half ? rowoffB : rowoffA;
                    const bool found = key != 0x7fffffff;
                    a.oscore[gp] = (int)(half ? knownB : knownA);
                    a.oqe[gp] = found ? (key & 2047) - rowoff : -2;
                    // MODE 6 marks "one row only" in the column output (SW_TE_UNIQUE): the mirror of such a pair shares its result
                    a.ote[gp] = found ? ((key >> 11) | ((MODE == 6 && nrows == 1) ? SW_TE_UNIQUE : 0)) : -2;
                }
            }
            return;
        }
#pragma unroll
        for (int half = 0; half < 2; half++) {
            int score = half ? (int)(best >> 16) : (int)(best & 0xffffu);
            int col = half ? colB : colA;
#pragma unroll
            for (int m = 1; m < G; m <<= 1) {
                const int os = __shfl_xor(score, m, 64), oc = __shfl_xor(col, m, 64);
                const bool take = os > score || (os == score && oc < col);
                score = take ? os : score;
                col = take ? oc : col;
            }
            int cnt = 0, minrow = 1 << 30;
            if constexpr (TRACK) {
#pragma unroll
                for (int r = R - 1; r >= 0; r--) {
                    const int rb = half ? (int)(rowbest[r] >> 16) : (int)(rowbest[r] & 0xffffu);
                    if (rb == score) { cnt++; minrow = g * R + r; }
                }
#pragma unroll
                for (int m = 1; m < G; m <<= 1) {
                    cnt += __shfl_xor(cnt, m, 64);
                    minrow = min(minrow, __shfl_xor(minrow, m, 64));
                }
            }
            const bool valid = half ? vB : true;
            const uint32_t gp = half ? gB : gA;
            if (g == 0 && valid) {
                a.oscore[gp] = score;
                if constexpr (TRACK) {
                    const int rowoff = half ? rowoffB : rowoffA;
                    int qe = -1, te = -1;
                    if (score > 0) { te = col; qe = (cnt == 1 && score < SW_PK_OVF) ? minrow - rowoff : -2; }
                    a.oqe[gp] = qe;
                    a.ote[gp] = te;
                }
            }
        }
    };

    start_slot();
    while (__builtin_amdgcn_ballot_w64(active) != 0) {
        if (active) {
            do_step(lst, false);
            do_step(lst + 1, true);
            lst += 2;
            if (lst >= nst) {
                finish_slot();
                start_slot();
            }
        }
    }
}

template <int MODE>
void launch_sw_pk_class_mode(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s) {
#define UC_SW_CASE(GG, RR)                                                                              \
    if (G == GG && R == RR) {                                                                           \
        constexpr int BW = (((RR + 3) / 4) | 1);                                                        \
        /* small workgroups share one LDS profile; G = 64 with R < 14: the wave-wide classes of sparse known-score plans (table 3 of uc_align.hip) */ \
        constexpr int NW = GG == 16 ? 2 : (GG == 32 ? 4 : (RR < 14 ? 2 : 8));                           \
        const size_t lds = (size_t)2 * SW_NLET * GG * BW * 4 + 16;                                      \
        static PerDeviceOnce once;                                                                      \
        if (lds > 64 * 1024)                                                                            \
            once([&] { (void)hipFuncSetAttribute((const void *)sw_pk_kernel<GG, RR, MODE, NW>,           \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }); \
        hipLaunchKernelGGL((sw_pk_kernel<GG, RR, MODE, NW>), dim3(n_tasks), dim3(NW * 64), lds, s, a);  \
        return;                                                                                         \
    }
    // even R: 32-row class steps for G=16, 64 for G=32, 128 for G=64 (row padding 11.7 % -> 6.2 % at C2, tools/geom_stats.py)
    UC_SW_CASE(16, 2) UC_SW_CASE(16, 4) UC_SW_CASE(16, 6) UC_SW_CASE(16, 8) UC_SW_CASE(16, 10) UC_SW_CASE(16, 12)
    UC_SW_CASE(16, 14) UC_SW_CASE(16, 16) UC_SW_CASE(16, 18) UC_SW_CASE(16, 20) UC_SW_CASE(16, 22) UC_SW_CASE(16, 24)
    UC_SW_CASE(32, 14) UC_SW_CASE(32, 16) UC_SW_CASE(32, 18) UC_SW_CASE(32, 20) UC_SW_CASE(32, 22) UC_SW_CASE(32, 24)
    UC_SW_CASE(64, 14) UC_SW_CASE(64, 16) UC_SW_CASE(64, 18) UC_SW_CASE(64, 20) UC_SW_CASE(64, 22) UC_SW_CASE(64, 24)
    UC_SW_CASE(64, 26) UC_SW_CASE(64, 28) UC_SW_CASE(64, 30) UC_SW_CASE(64, 32)   // ~6R + 60 live registers: R = 32 still fits 256 VGPRs
    if constexpr (MODE == 4 || MODE == 6) {   // sparse plans only exist for the known-score passes
        UC_SW_CASE(64, 2) UC_SW_CASE(64, 4) UC_SW_CASE(64, 6) UC_SW_CASE(64, 8) UC_SW_CASE(64, 10) UC_SW_CASE(64, 12)
    }
#undef UC_SW_CASE
    fprintf(stderr, "unicore-cluster: no packed SW kernel for class (G=%d, R=%d)\n", G, R);
    abort();
}

// forces this translation unit's code object onto the current device (HIP loads a module lazily at the first use of one of its kernels): see preload_modules()
template <int MODE>
void preload_sw_pk_mode() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)sw_pk_kernel<64, 14, MODE, 8>);
}

}  // namespace uc
