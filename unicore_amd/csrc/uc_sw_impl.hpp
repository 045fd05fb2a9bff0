// uc_sw_impl.hpp — stage E5: gapped 3Di+AA Smith-Waterman as a wavefront-primitive anti-diagonal DP
// kernel for gfx950 (wave64).  Replaces Foldseek's structurealign / SSW inside `foldseek cluster`
// (call site /root/reference/src/modules/cluster.rs:45-56; algorithm SURVEY.md A.3; spec UC-1 E5).
//
// Design (integer VALU DP — no MFMA: this is not a contraction):
//  * A group of G lanes (G = 16/32/64, 64/G alignments per wave) sweeps ONE alignment as a systolic
//    anti-diagonal: lane g owns R consecutive query rows, at step st it computes its R cells of target
//    column j = st - g.  Column results flow to lane g+1 through DPP row_shr:1 (G=16: one DPP row per
//    alignment) or ds_bpermute (G=32/64); no DP state ever leaves registers.
//  * The four alignments of a wave share one query: the workgroup stages a per-query PROFILE in LDS
//    once (prof[track][letter][row] as biased bytes, 4 rows per dword, row blocks padded to an odd dword
//    stride so the 16 lanes of a group hit 16 distinct banks) and then streams all of that query's
//    targets through it, so the inner loop does R/4 ds_read_b32 per track per step, one packed add and
//    one xor to form four signed (S3+SA+open) bytes, and v_dot4_i32_i8 to fold byte r into H(i-1,j-1).
//  * DP state per row is (T = H - open, E>=0).  E and F live in the floored domain (max(.,0)), which
//    is exact for local alignment and lets v_sub_u32 clamp + v_max3_i32 form H without a zero-compare.
//  * End tracking (tie-break: smallest tEnd, then smallest qEnd) costs one v_lshl_or per cell: the
//    row id rides in the low 5 bits of a keyed copy of H, the column comes from the step counter.
//  * MODE 0 forward, MODE 1 reversed query (score only), MODE 2 start pass on the reversed prefixes
//    (reversed query profile with the rows beyond qEnd masked to PAD, target read backwards from tEnd).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "uc_device.h"

namespace uc {

constexpr int SW_NLET = 22;            // 21 letters + PAD(21)
constexpr uint32_t SW_PADPACK = 21u | (21u << 8);

// waves per workgroup: every wave streams targets through the ONE LDS profile of the task's query, so the
// classes with big profiles (1 or 2 workgroups per CU by LDS) use more waves to keep the SIMDs occupied
constexpr int sw_waves_per_group(int G, int R) { return G < 64 ? 4 : (R > 28 ? 12 : 8); }

template <int G>
__device__ __forceinline__ int shift_from_prev_lane(int v, int boundary, int g) {
    if constexpr (G == 16) {
        // DPP row_shr:1 — lane 0 of each 16-lane row keeps `old` (= boundary)
        return __builtin_amdgcn_update_dpp(boundary, v, 0x111, 0xf, 0xf, false);
    } else {
        // DPP wave_shr:1 crosses the 16-lane rows; lane 0 of the wave keeps `old`, lane 32 of a
        // G=32 group is patched with the boundary
        const int r = __builtin_amdgcn_update_dpp(boundary, v, 0x138, 0xf, 0xf, false);
        return (G == 64 || g != 0) ? r : boundary;
    }
}

// boundary 0: DPP bound_ctrl writes 0 into the lanes without a source, so no register has to be preset
template <int G>
__device__ __forceinline__ int shift_from_prev_lane_zero(int v, int g) {
    if constexpr (G == 16) {
        return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    } else {
        const int r = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
        return (G == 64 || g != 0) ? r : 0;
    }
}

// NW = waves per workgroup: all of them stream targets through the one LDS profile of the task's query
template <int G, int R, int MODE, int NW>
__global__ void __launch_bounds__(NW * 64) sw_group_kernel(const SwArgs a) {
    // MODE 3 (traceback statistics): forward DP on the box [qs..qe] x [ts..te] carrying (alignment length,
    // identities) along the chosen predecessor of every state — the numbers the traceback of spec UC-1 E6 yields
    constexpr bool TB = MODE == 3;
    constexpr bool TRACK = MODE == 0 || MODE == 2, MASK = MODE == 2 || TB, REVQ = MODE == 1 || MODE == 2, REVT = MODE == 2;
    constexpr int RW = R / 4, BW = RW | 1, RSW = G * BW, GPW = 64 / G, NT = NW * 64;
    static_assert(R % 4 == 0 && R <= 32, "R must be a multiple of 4, <= 32");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *P3 = lds, *PA = lds + SW_NLET * RSW;

    const SwTask task = a.tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane % G, grp = lane / G;
    const uint32_t qoff = a.db.off[task.q];
    const int lq = (int)a.db.len[task.q];
    const int open = a.open, ext = a.ext;

    // ---- stage the query profile in LDS (biased bytes: b3 = S3+64, bA = SA+64+open; PAD = 0) ----
    for (int idx = tid; idx < SW_NLET * G * RW; idx += NT) {
        const int c = idx / (G * RW), rem = idx % (G * RW), gg = rem / RW, k = rem % RW;
        uint32_t w3 = 0, wa = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int row = gg * R + 4 * k + b;
            if (row < lq && c < 21) {
                const int qi = REVQ ? lq - 1 - row : row;
                const int q3 = a.db.s3[qoff + qi], qa = a.db.sa[qoff + qi];
                w3 |= (uint32_t)(a.db.S3[q3 * 21 + c] + 64) << (8 * b);
                wa |= (uint32_t)(a.db.SA[qa * 21 + c] + 64 + open) << (8 * b);
            }
        }
        P3[c * RSW + gg * BW + k] = w3;
        PA[c * RSW + gg * BW + k] = wa;
    }
    __syncthreads();
    uint32_t qaw[TB ? RW : 1];   // TB: AA letters of this lane's rows (0xFF beyond the query)
    if constexpr (TB) {
#pragma unroll
        for (int k = 0; k < RW; k++) {
            uint32_t w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int row = g * R + 4 * k + b;
                w |= (row < lq ? (uint32_t)a.db.sa[qoff + row] : 0xFFu) << (8 * b);
            }
            qaw[k] = w;
        }
    }

    for (uint32_t pb = (uint32_t)wave * GPW; pb < task.count; pb += NW * GPW) {
        const bool pvalid = pb + grp < task.count;
        const uint32_t gp = task.begin + (pvalid ? pb + grp : 0);
        const uint32_t t = a.pt[gp];
        const uint32_t toff = a.db.off[t];
        const int tstart = TB ? a.pts[gp] : 0;
        int tlen = TB ? a.pte[gp] - tstart + 1 : (REVT ? a.pte[gp] + 1 : (int)a.db.len[t]);
        if (!pvalid) tlen = 0;
        const int tlast = tlen - 1;
        int rowoff = 0;
        uint32_t msk[RW];
        if constexpr (MASK) {
            rowoff = TB ? a.pqs[gp] : lq - 1 - a.pqe[gp];
#pragma unroll
            for (int k = 0; k < RW; k++) {
                uint32_t m = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) m |= (g * R + 4 * k + b >= rowoff ? 0xFFu : 0u) << (8 * b);
                msk[k] = m;
            }
        }
        int maxlen = 0;
#pragma unroll
        for (int i = 0; i < GPW; i++) maxlen = max(maxlen, __builtin_amdgcn_readlane(tlen, i * G));
        const int nsteps = maxlen > 0 ? maxlen + G - 1 : 0;

        int T[R];
        uint32_t E[R];
#pragma unroll
        for (int r = 0; r < R; r++) { T[r] = -open; E[r] = 0; }
        uint32_t best = 0;       // TRACK: keyed (score<<5 | 31-r); else plain score
        int bestcol = -1;
        int Tlast = -open, prevTup = -open;
        uint32_t fout = 0;
        // TB state: packed (aln_len << 16 | idents) of the path into H / E per row, F flowing down the column
        uint32_t Hp[TB ? R : 1], Ep[TB ? R : 1];
        uint32_t HpLast = 0, prevHpUp = 0, FpOut = 0, cap = 0;
        const int qe_row = TB ? a.pqe[gp] : 0;
        if constexpr (TB) {
#pragma unroll
            for (int r = 0; r < R; r++) { Hp[r] = 0; Ep[r] = 0; }
        }

        // target letters: branch-free byte loads (clamped index, uniform within the group), combined one full
        // step after they were issued so that no step waits on global-memory latency
        struct RawLetter { uint32_t c3, ca; };
        auto issue_letter = [&](int st) -> RawLetter {
            const int i = max(min(st, tlen - 1), 0);
            const uint32_t p = toff + (uint32_t)(REVT ? max(tlast - i, 0) : tstart + i);
            RawLetter r;
            r.c3 = a.db.s3[p]; r.ca = a.db.sa[p];
            return r;
        };
        auto pack_letter = [&](const RawLetter &r, int st) -> uint32_t { return st < tlen ? (r.c3 | (r.ca << 8)) : SW_PADPACK; };
        uint32_t c1 = pack_letter(issue_letter(1), 1);
        RawLetter r2 = issue_letter(2);
        uint32_t cin = (uint32_t)shift_from_prev_lane<G>((int)SW_PADPACK, (int)pack_letter(issue_letter(0), 0), g);
        uint32_t n3[RW], na[RW];
        {
            const uint32_t *p3 = P3 + (cin & 0xff) * RSW + g * BW, *pa = PA + (cin >> 8) * RSW + g * BW;
#pragma unroll
            for (int k = 0; k < RW; k++) { n3[k] = p3[k]; na[k] = pa[k]; }
        }

        // two steps per trip (an extra all-PAD step is harmless) so that the rotating (T, old T) register
        // pairs need no copies at the back edge
        auto do_step = [&](const int st) __attribute__((always_inline)) {
            uint32_t ps[RW];
            [[maybe_unused]] const uint32_t cin_cur = cin;   // letters of the column this step computes
#pragma unroll
            for (int k = 0; k < RW; k++) {
                uint32_t s = n3[k] + na[k];
                if constexpr (MASK) s &= msk[k];
                ps[k] = s ^ 0x80808080u;
            }
            // next column's letters and profile words
            cin = (uint32_t)shift_from_prev_lane<G>((int)cin, (int)c1, g);
            c1 = pack_letter(r2, st + 2);
            r2 = issue_letter(st + 3);
            {
                const uint32_t *p3 = P3 + (cin & 0xff) * RSW + g * BW, *pa = PA + (cin >> 8) * RSW + g * BW;
#pragma unroll
                for (int k = 0; k < RW; k++) { n3[k] = p3[k]; na[k] = pa[k]; }
            }
            const int Tup = shift_from_prev_lane<G>(Tlast, -open, g);
            uint32_t f = (uint32_t)shift_from_prev_lane<G>((int)fout, 0, g);
            int diagT = prevTup;
            uint32_t colmax = 0;
            [[maybe_unused]] uint32_t HpUp = 0, fp = 0, dHp = 0;
            [[maybe_unused]] const uint32_t ca_col = cin_cur >> 8;
            if constexpr (TB) {
                HpUp = (uint32_t)shift_from_prev_lane<G>((int)HpLast, 0, g);
                fp = (uint32_t)shift_from_prev_lane<G>((int)FpOut, 0, g);
                dHp = prevHpUp;
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int x = __builtin_amdgcn_sdot4((int)ps[r >> 2], 1 << (8 * (r & 3)), diagT, false);
                const uint32_t esub = __builtin_elementwise_sub_sat(E[r], (uint32_t)ext);
                const int e = max((int)esub, T[r]);
                const int h = max(max(x, e), (int)f);
                if constexpr (TB) {
                    // predecessor preference of the traceback: diagonal, then F (gap in the target), then E;
                    // a gap state prefers leaving the gap (open) over staying in it
                    const uint32_t ep = T[r] >= (int)esub ? Hp[r] + a.tb_open : Ep[r] + a.tb_ext;
                    const uint32_t ident = ((qaw[r >> 2] >> (8 * (r & 3))) & 0xffu) == ca_col ? a.tb_ident : 0u;
                    // bit 31 marks a traceback step that had to choose between the two gap directions (F preferred over E):
                    // the mirrored pair (t,q) would choose the other one, so only tie-free statistics may be shared with it
                    const uint32_t tie = (x != h && (int)f == h && e == h) ? 0x80000000u : 0u;
                    const uint32_t hp = h == 0 ? 0u : (x == h ? dHp + a.tb_diag + ident : ((int)f == h ? (fp | tie) : ep));
                    dHp = Hp[r];
                    Hp[r] = hp;
                    Ep[r] = ep;
                    const uint32_t fsub_ = __builtin_elementwise_sub_sat(f, (uint32_t)ext);
                    fp = h - open >= (int)fsub_ ? hp + a.tb_open : fp + a.tb_ext;
                }
                diagT = T[r];
                T[r] = h - open;
                E[r] = (uint32_t)e;
                f = (uint32_t)max((int)__builtin_elementwise_sub_sat(f, (uint32_t)ext), T[r]);
                if constexpr (TRACK) colmax = max(colmax, ((uint32_t)h << 5) | (uint32_t)(31 - r));
                else colmax = max(colmax, (uint32_t)h);
            }
            if constexpr (TRACK) {
                const bool upd = colmax > (best | 31u);
                best = upd ? colmax : best;
                bestcol = upd ? st - g : bestcol;
            } else {
                best = max(best, colmax);
            }
            if constexpr (TB) {
                if (st - g == tlen - 1) {          // the box's last column: keep the pack of row qe
#pragma unroll
                    for (int r = 0; r < R; r++)
                        if (g * R + r == qe_row) cap = Hp[r];
                }
                HpLast = Hp[R - 1];
                FpOut = fp;
                prevHpUp = HpUp;
            }
            Tlast = T[R - 1];
            fout = f;
            prevTup = Tup;
        };
        for (int st = 0; st < nsteps; st += 2) {
            do_step(st);
            do_step(st + 1);
        }

        // ---- reduce over the G lanes of the group: (score desc, col asc, row asc) ----
        int score = TRACK ? (int)(best >> 5) : (int)best;
        int row = TRACK ? g * R + (31 - (int)(best & 31u)) : 0;
        int col = bestcol;
#pragma unroll
        for (int m = 1; m < G; m <<= 1) {
            const int os = __shfl_xor(score, m, 64), oc = __shfl_xor(col, m, 64), orow = __shfl_xor(row, m, 64);
            const bool take = os > score || (os == score && (oc < col || (oc == col && orow < row)));
            score = take ? os : score;
            col = take ? oc : col;
            row = take ? orow : row;
        }
        if constexpr (TB) {          // exactly one lane of the group holds row qe
#pragma unroll
            for (int m = 1; m < G; m <<= 1) cap |= (uint32_t)__shfl_xor((int)cap, m, 64);
            score = (int)cap;
        }
        if (g == 0 && pvalid) {
            a.oscore[gp] = score;
            if constexpr (TRACK) {
                a.oqe[gp] = score > 0 ? row - rowoff : -1;
                a.ote[gp] = score > 0 ? col : -1;
            }
        }
    }
}

template <int MODE>
void launch_sw_class_mode(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s) {
#define UC_SW_CASE(GG, RR)                                                                              \
    if (G == GG && R == RR) {                                                                           \
        constexpr int BW = ((RR / 4) | 1);                                                              \
        constexpr int NW = MODE == 3 ? (GG == 64 ? 8 : 4) : sw_waves_per_group(GG, RR);               \
        const size_t lds = (size_t)2 * SW_NLET * GG * BW * 4;                                           \
        static PerDeviceOnce once;                                                                      \
        if (lds > 64 * 1024)                                                                            \
            once([&] { (void)hipFuncSetAttribute((const void *)sw_group_kernel<GG, RR, MODE, NW>,           \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }); \
        hipLaunchKernelGGL((sw_group_kernel<GG, RR, MODE, NW>), dim3(n_tasks), dim3(NW * 64), lds, s, a); \
        return;                                                                                         \
    }
    UC_SW_CASE(16, 4) UC_SW_CASE(16, 8) UC_SW_CASE(16, 12) UC_SW_CASE(16, 16)
    UC_SW_CASE(16, 20) UC_SW_CASE(16, 24) UC_SW_CASE(16, 28) UC_SW_CASE(16, 32)
    UC_SW_CASE(32, 12) UC_SW_CASE(32, 16) UC_SW_CASE(32, 20) UC_SW_CASE(32, 24) UC_SW_CASE(32, 28) UC_SW_CASE(32, 32)
    UC_SW_CASE(64, 12) UC_SW_CASE(64, 16) UC_SW_CASE(64, 20) UC_SW_CASE(64, 24) UC_SW_CASE(64, 28) UC_SW_CASE(64, 32)
#undef UC_SW_CASE
    fprintf(stderr, "unicore-cluster: no SW kernel for class (G=%d, R=%d)\n", G, R);
    abort();
}

}  // namespace uc
