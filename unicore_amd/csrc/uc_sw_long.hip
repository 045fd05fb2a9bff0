// uc_sw_long.hip — gapped DP (stage E5) for queries beyond the largest systolic class (> 2048 rows).
//
// Same wavefront design as sw_group_kernel (uc_sw_impl.hpp), int32 arithmetic, G = 64 lanes x R = 32 rows, but the
// query is swept in ROW BLOCKS of 2048 rows: block b of a pair starts from the boundary that block b-1 left behind —
// for every target column the H value of the block's last row (kept as T = H - open) and the F value leaving it.
// Lane 63 writes the boundary of column c at step c + 63, lane 0 of the next block reads it at step c, so one pair
// of arrays per alignment is updated in place.  A workgroup is one task = one query + up to LONG_NW of its pairs,
// one pair per wave; the LDS profile holds the current row block and is rebuilt (workgroup barrier) per block.
// Replaces the one-lane-per-pair fallback for MODE 0 / 1 / 2 (that one needed ~0.8 s per pass for 64 pairs of
// 2500 x 3000 residues and grows with Lq x Lt per LANE; it remains for the traceback statistics, MODE 3).
// Spec: the same UC-1 recurrence and tie-break as every other SW kernel (smallest tEnd, then smallest qEnd).
#include "uc_sw_impl.hpp"

namespace uc {

constexpr int LONG_G = 64, LONG_R = 32, LONG_ROWS = LONG_G * LONG_R;
constexpr int LONG_NW = (int)SW_LONG_TASK_PAIRS;   // waves (= pairs) per workgroup; uc_align.hip cuts the long-query tasks to this size

template <int MODE>
__global__ void __launch_bounds__(LONG_NW * 64) sw_long_kernel(const SwArgs a, uint32_t pair_base, int32_t *work, uint32_t stride) {
    constexpr int G = LONG_G, R = LONG_R, NW = LONG_NW;
    constexpr bool TRACK = MODE == 0 || MODE == 2, MASK = MODE == 2, REVQ = MODE == 1 || MODE == 2, REVT = MODE == 2;
    constexpr int RW = R / 4, BW = RW | 1, RSW = G * BW, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *P3 = lds, *PA = lds + SW_NLET * RSW;

    const SwTask task = a.tasks[blockIdx.x];
    const int tid = threadIdx.x, g = tid & 63, wave = tid >> 6;
    const uint32_t qoff = a.db.off[task.q];
    const int lq = (int)a.db.len[task.q];
    const int open = a.open, ext = a.ext;
    const int nblk = (lq + LONG_ROWS - 1) / LONG_ROWS;

    // this wave's pair
    const bool pvalid = (uint32_t)wave < task.count;
    const uint32_t gp = task.begin + (pvalid ? (uint32_t)wave : 0u);
    const uint32_t t = a.pt[gp];
    const uint32_t toff = a.db.off[t];
    int tlen = REVT ? a.pte[gp] + 1 : (int)a.db.len[t];
    if (!pvalid) tlen = 0;
    const int tlast = tlen - 1;
    const int rowoff = MASK ? lq - 1 - a.pqe[gp] : 0;       // MODE 2: rows before rowoff are PAD
    const int blk0 = rowoff / LONG_ROWS;                     // ... so whole blocks before it leave the initial boundary
    int32_t *bT = work + (size_t)(gp - pair_base) * (2 * (size_t)stride);
    int32_t *bF = bT + stride;
    const int nsteps = tlen > 0 ? ((tlen + G - 1 + 1) & ~1) : 0;

    int bscore = 0, bcol = -1, brow = -1;                    // running optimum over the blocks (wave-uniform)

    for (int blk = 0; blk < nblk; blk++) {
        const int row0 = blk * LONG_ROWS;
        __syncthreads();                                     // every wave is done with the previous block's profile
        for (int idx = tid; idx < SW_NLET * G * RW; idx += NT) {
            const int c = idx / (G * RW), rem = idx % (G * RW), gg = rem / RW, k = rem % RW;
            uint32_t w3 = 0, wa = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int row = row0 + gg * R + 4 * k + b;
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
        __syncthreads();                                     // also orders the boundary stores of block blk-1 before the loads below
        if (!pvalid || blk < blk0) continue;
        const bool first = blk == blk0, last = blk == nblk - 1;

        uint32_t msk[MASK ? RW : 1];
        if constexpr (MASK) {
#pragma unroll
            for (int k = 0; k < RW; k++) {
                uint32_t m = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) m |= (row0 + g * R + 4 * k + b >= rowoff ? 0xFFu : 0u) << (8 * b);
                msk[k] = m;
            }
        }
        int T[R];
        uint32_t E[R];
#pragma unroll
        for (int r = 0; r < R; r++) { T[r] = -open; E[r] = 0; }
        uint32_t best = 0;       // TRACK: keyed (score<<5 | 31-r); else plain score
        int bestcol = -1;
        int Tlast = -open, prevTup = -open;
        uint32_t fout = 0;

        struct RawLetter { uint32_t c3, ca; };
        auto issue_letter = [&](int st) -> RawLetter {
            const int i = max(min(st, tlen - 1), 0);
            const uint32_t p = toff + (uint32_t)(REVT ? max(tlast - i, 0) : i);
            RawLetter r;
            r.c3 = a.db.s3[p]; r.ca = a.db.sa[p];
            return r;
        };
        auto pack_letter = [&](const RawLetter &r, int st) -> uint32_t { return st < tlen ? (r.c3 | (r.ca << 8)) : SW_PADPACK; };
        // boundary of the row above this block for column st (agent-scope loads: written by lane 63 through L2)
        struct Bnd { int t; uint32_t f; };
        auto issue_bnd = [&](int st) -> Bnd {
            Bnd b = {-open, 0u};
            if (!first) {
                const int i = max(min(st, tlen - 1), 0);
                b.t = __hip_atomic_load(bT + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.f = (uint32_t)__hip_atomic_load(bF + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return b;
        };
        uint32_t c1 = pack_letter(issue_letter(1), 1);
        RawLetter r2 = issue_letter(2);
        uint32_t cin = (uint32_t)shift_from_prev_lane<G>((int)SW_PADPACK, (int)pack_letter(issue_letter(0), 0), g);
        Bnd b0 = issue_bnd(0), b1 = issue_bnd(1);
        uint32_t n3[RW], na[RW];
        {
            const uint32_t *p3 = P3 + (cin & 0xff) * RSW + g * BW, *pa = PA + (cin >> 8) * RSW + g * BW;
#pragma unroll
            for (int k = 0; k < RW; k++) { n3[k] = p3[k]; na[k] = pa[k]; }
        }

        auto do_step = [&](const int st) __attribute__((always_inline)) {
            uint32_t ps[RW];
#pragma unroll
            for (int k = 0; k < RW; k++) {
                uint32_t s = n3[k] + na[k];
                if constexpr (MASK) s &= msk[k];
                ps[k] = s ^ 0x80808080u;
            }
            cin = (uint32_t)shift_from_prev_lane<G>((int)cin, (int)c1, g);
            c1 = pack_letter(r2, st + 2);
            r2 = issue_letter(st + 3);
            {
                const uint32_t *p3 = P3 + (cin & 0xff) * RSW + g * BW, *pa = PA + (cin >> 8) * RSW + g * BW;
#pragma unroll
                for (int k = 0; k < RW; k++) { n3[k] = p3[k]; na[k] = pa[k]; }
            }
            // lane 0 takes the row above from the boundary of column st (PAD columns beyond the target: the initial values)
            const Bnd bc = st < tlen ? b0 : Bnd{-open, 0u};
            b0 = b1;
            b1 = issue_bnd(st + 2);
            const int Tup = shift_from_prev_lane<G>(Tlast, bc.t, g);
            uint32_t f = (uint32_t)shift_from_prev_lane<G>((int)fout, (int)bc.f, g);
            int diagT = prevTup;
            uint32_t colmax = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int x = __builtin_amdgcn_sdot4((int)ps[r >> 2], 1 << (8 * (r & 3)), diagT, false);
                const uint32_t esub = __builtin_elementwise_sub_sat(E[r], (uint32_t)ext);
                const int e = max((int)esub, T[r]);
                const int h = max(max(x, e), (int)f);
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
            Tlast = T[R - 1];
            fout = f;
            prevTup = Tup;
            if (!last && g == G - 1) {      // boundary for the next block: this lane's last row, column st - 63
                const int col = st - (G - 1);
                if (col >= 0 && col < tlen) {
                    __hip_atomic_store(bT + col, Tlast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(bF + col, (int32_t)fout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };
        for (int st = 0; st < nsteps; st += 2) {
            do_step(st);
            do_step(st + 1);
        }

        // reduce over the 64 lanes: (score desc, col asc, row asc), then fold into the running optimum of the pair
        int score = TRACK ? (int)(best >> 5) : (int)best;
        int row = TRACK ? row0 + g * R + (31 - (int)(best & 31u)) : 0;
        int col = bestcol;
#pragma unroll
        for (int m = 1; m < G; m <<= 1) {
            const int os = __shfl_xor(score, m, 64), oc = __shfl_xor(col, m, 64), orow = __shfl_xor(row, m, 64);
            const bool take = os > score || (os == score && (oc < col || (oc == col && orow < row)));
            score = take ? os : score;
            col = take ? oc : col;
            row = take ? orow : row;
        }
        const bool take = score > bscore || (TRACK && score == bscore && score > 0 && (col < bcol || (col == bcol && row < brow)));
        if (take) { bscore = score; bcol = col; brow = row; }
    }
    if (g == 0 && pvalid) {
        a.oscore[gp] = bscore;
        if constexpr (TRACK) {
            a.oqe[gp] = bscore > 0 ? brow - rowoff : -1;
            a.ote[gp] = bscore > 0 ? bcol : -1;
        }
    }
}

size_t sw_long_work_ints(uint32_t n_pairs, uint32_t max_len, uint32_t *stride) {
    *stride = (max_len + 63u) & ~63u;
    return (size_t)n_pairs * 2 * (size_t)*stride;
}

void launch_sw_long(int mode, const SwArgs &a, uint32_t n_tasks, uint32_t pair_base, int32_t *work, uint32_t stride, hipStream_t s) {
    if (n_tasks == 0) return;
    constexpr int BW = ((LONG_R / 4) | 1);
    const size_t lds = (size_t)2 * SW_NLET * LONG_G * BW * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)sw_long_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)sw_long_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)sw_long_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const dim3 grid(n_tasks), block(LONG_NW * 64);
    if (mode == 0) hipLaunchKernelGGL(sw_long_kernel<0>, grid, block, lds, s, a, pair_base, work, stride);
    else if (mode == 1) hipLaunchKernelGGL(sw_long_kernel<1>, grid, block, lds, s, a, pair_base, work, stride);
    else hipLaunchKernelGGL(sw_long_kernel<2>, grid, block, lds, s, a, pair_base, work, stride);
}

}  // namespace uc
