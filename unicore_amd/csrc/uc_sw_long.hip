// uc_sw_long.hip — gapped DP (stage E5 / E6 statistics) for queries beyond the largest systolic class (> 2048 rows).
//
// Same wavefront design as sw_group_kernel (uc_sw_impl.hpp), int32 arithmetic, G = 64 lanes x R rows, but the query
// is swept in ROW BLOCKS of 64 R rows: block b of a pair starts from the boundary that block b-1 left behind — for
// every target column the H value of the block's last row (kept as T = H - open) and the F value leaving it (MODE 3:
// also the traceback packs travelling with them).  Lane 63 writes the boundary of column c at step c + 63, lane 0 of
// the next block reads it at step c, so one set of arrays per alignment is updated in place.  A workgroup is one
// task = one query + up to SW_LONG_TASK_PAIRS of its pairs, one pair per wave; the LDS profile holds the current row
// block and is rebuilt (workgroup barrier) per block.  Blocks that cannot matter are skipped: the masked prefix of the
// start pass / of the traceback box, and in MODE 3 everything below the box.
// It replaced a one-lane-per-pair fallback that needed ~0.8 s per pass for 64 pairs of 2500 x 3000 residues (and grew
// with Lq x Lt per LANE).  Spec: the UC-1 recurrence and tie-break of every other SW kernel (smallest tEnd, then
// smallest qEnd); MODE 3 = the traceback statistics of sw_group_kernel<.., 3> (see there for the pack semantics).
#include "uc_sw_impl.hpp"

namespace uc {

constexpr int LONG_G = 64;
constexpr int LONG_NW = (int)SW_LONG_TASK_PAIRS;   // waves (= pairs) per workgroup; uc_align.hip cuts the long-query tasks to this size
constexpr int long_rows_per_lane(int mode) { return mode == 3 ? 16 : 32; }   // MODE 3 carries two more register arrays

template <int MODE>
__global__ void __launch_bounds__(LONG_NW * 64) sw_long_kernel(const SwArgs a, uint32_t pair_base, int32_t *work, uint32_t stride) {
    constexpr int G = LONG_G, R = long_rows_per_lane(MODE), NW = LONG_NW, ROWS = G * R;
    constexpr bool TB = MODE == 3;
    constexpr bool TRACK = MODE == 0 || MODE == 2, MASK = MODE == 2 || TB, REVQ = MODE == 1 || MODE == 2, REVT = MODE == 2;
    constexpr int RW = R / 4, BW = RW | 1, RSW = G * BW, NT = NW * 64, NARR = TB ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *P3 = lds, *PA = lds + SW_NLET * RSW;

    const SwTask task = a.tasks[blockIdx.x];
    const int tid = threadIdx.x, g = tid & 63, wave = tid >> 6;
    const uint32_t qoff = a.db.off[task.q];
    const int lq = (int)a.db.len[task.q];
    const int open = a.open, ext = a.ext;

    // this wave's pair
    const bool pvalid = (uint32_t)wave < task.count;
    const uint32_t gp = task.begin + (pvalid ? (uint32_t)wave : 0u);
    const uint32_t t = a.pt[gp];
    const uint32_t toff = a.db.off[t];
    const int tstart = TB ? a.pts[gp] : 0;
    int tlen = TB ? a.pte[gp] - tstart + 1 : (REVT ? a.pte[gp] + 1 : (int)a.db.len[t]);
    if (!pvalid) tlen = 0;
    const int tlast = tlen - 1;
    const int rowoff = TB ? a.pqs[gp] : (MASK ? lq - 1 - a.pqe[gp] : 0);   // rows before rowoff are PAD ...
    const int blk0 = rowoff / ROWS;                                        // ... so whole blocks before it leave the initial boundary
    const int qe_row = TB ? a.pqe[gp] : 0;                                 // MODE 3: the box ends in this row
    // blocks of the task: the whole query, MODE 3: up to the deepest box of its pairs (workgroup-uniform)
    int nblk = (lq + ROWS - 1) / ROWS;
    if constexpr (TB) {
        __shared__ int s_last;
        if (tid == 0) s_last = 0;
        __syncthreads();
        if (pvalid && g == 0) atomicMax(&s_last, qe_row / ROWS);
        __syncthreads();
        nblk = s_last + 1;
    }
    const int blk_end = TB ? qe_row / ROWS : nblk - 1;                     // last block this pair needs
    int32_t *bT = work + (size_t)(gp - pair_base) * (NARR * (size_t)stride);
    int32_t *bF = bT + stride;
    [[maybe_unused]] int32_t *bHp = bF + stride, *bFp = bF + 2 * (size_t)stride;
    const int nsteps = tlen > 0 ? ((tlen + G - 1 + 1) & ~1) : 0;

    int bscore = 0, bcol = -1, brow = -1;                    // running optimum over the blocks (wave-uniform)
    [[maybe_unused]] uint32_t cap = 0;                       // MODE 3: the pack of cell (qe, te)

    for (int blk = 0; blk < nblk; blk++) {
        const int row0 = blk * ROWS;
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
        if (!pvalid || blk < blk0 || blk > blk_end) continue;
        const bool first = blk == blk0, last = blk == blk_end;

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
        uint32_t qaw[TB ? RW : 1];   // TB: AA letters of this lane's rows (0xFF beyond the query)
        if constexpr (TB) {
#pragma unroll
            for (int k = 0; k < RW; k++) {
                uint32_t w = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int row = row0 + g * R + 4 * k + b;
                    w |= (row < lq ? (uint32_t)a.db.sa[qoff + row] : 0xFFu) << (8 * b);
                }
                qaw[k] = w;
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
        uint32_t Hp[TB ? R : 1], Ep[TB ? R : 1];
        [[maybe_unused]] uint32_t HpLast = 0, prevHpUp = 0, FpOut = 0;
        if constexpr (TB) {
#pragma unroll
            for (int r = 0; r < R; r++) { Hp[r] = 0; Ep[r] = 0; }
        }

        struct RawLetter { uint32_t c3, ca; };
        auto issue_letter = [&](int st) -> RawLetter {
            const int i = max(min(st, tlen - 1), 0);
            const uint32_t p = toff + (uint32_t)(REVT ? max(tlast - i, 0) : tstart + i);
            RawLetter r;
            r.c3 = a.db.s3[p]; r.ca = a.db.sa[p];
            return r;
        };
        auto pack_letter = [&](const RawLetter &r, int st) -> uint32_t { return st < tlen ? (r.c3 | (r.ca << 8)) : SW_PADPACK; };
        // boundary of the row above this block for column st (agent-scope loads: written by lane 63 through L2)
        struct Bnd { int t; uint32_t f, hp, fp; };
        auto issue_bnd = [&](int st) -> Bnd {
            Bnd b = {-open, 0u, 0u, 0u};
            if (!first) {
                const int i = max(min(st, tlen - 1), 0);
                b.t = __hip_atomic_load(bT + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.f = (uint32_t)__hip_atomic_load(bF + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if constexpr (TB) {
                    b.hp = (uint32_t)__hip_atomic_load(bHp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    b.fp = (uint32_t)__hip_atomic_load(bFp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
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
            [[maybe_unused]] const uint32_t cin_cur = cin;   // letters of the column this step computes
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
            const Bnd bc = st < tlen ? b0 : Bnd{-open, 0u, 0u, 0u};
            b0 = b1;
            b1 = issue_bnd(st + 2);
            const int Tup = shift_from_prev_lane<G>(Tlast, bc.t, g);
            uint32_t f = (uint32_t)shift_from_prev_lane<G>((int)fout, (int)bc.f, g);
            int diagT = prevTup;
            uint32_t colmax = 0;
            [[maybe_unused]] uint32_t HpUp = 0, fp = 0, dHp = 0;
            [[maybe_unused]] const uint32_t ca_col = cin_cur >> 8;
            if constexpr (TB) {
                HpUp = (uint32_t)shift_from_prev_lane<G>((int)HpLast, (int)bc.hp, g);
                fp = (uint32_t)shift_from_prev_lane<G>((int)FpOut, (int)bc.fp, g);
                dHp = prevHpUp;
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int x = __builtin_amdgcn_sdot4((int)ps[r >> 2], 1 << (8 * (r & 3)), diagT, false);
                const uint32_t esub = __builtin_elementwise_sub_sat(E[r], (uint32_t)ext);
                const int e = max((int)esub, T[r]);
                const int h = max(max(x, e), (int)f);
                if constexpr (TB) {   // see sw_group_kernel: predecessor preference diagonal > F > E, gap states prefer to open
                    const uint32_t ep = T[r] >= (int)esub ? Hp[r] + a.tb_open : Ep[r] + a.tb_ext;
                    const uint32_t ident = ((qaw[r >> 2] >> (8 * (r & 3))) & 0xffu) == ca_col ? a.tb_ident : 0u;
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
                        if (row0 + g * R + r == qe_row) cap = Hp[r];
                }
                HpLast = Hp[R - 1];
                FpOut = fp;
                prevHpUp = HpUp;
            }
            Tlast = T[R - 1];
            fout = f;
            prevTup = Tup;
            if (!last && g == G - 1) {      // boundary for the next block: this lane's last row, column st - 63
                const int col = st - (G - 1);
                if (col >= 0 && col < tlen) {
                    __hip_atomic_store(bT + col, Tlast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(bF + col, (int32_t)fout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if constexpr (TB) {
                        __hip_atomic_store(bHp + col, (int32_t)HpLast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(bFp + col, (int32_t)FpOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        };
        for (int st = 0; st < nsteps; st += 2) {
            do_step(st);
            do_step(st + 1);
        }

        if constexpr (!TB) {
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
    }
    if constexpr (TB) {          // exactly one lane of one block holds row qe
#pragma unroll
        for (int m = 1; m < G; m <<= 1) cap |= (uint32_t)__shfl_xor((int)cap, m, 64);
        bscore = (int)cap;
    }
    if (g == 0 && pvalid) {
        a.oscore[gp] = bscore;
        if constexpr (TRACK) {
            a.oqe[gp] = bscore > 0 ? brow - rowoff : -1;
            a.ote[gp] = bscore > 0 ? bcol : -1;
        }
    }
}

size_t sw_long_work_ints(int mode, uint32_t n_pairs, uint32_t max_len, uint32_t *stride) {
    *stride = (max_len + 63u) & ~63u;
    return (size_t)n_pairs * (mode == 3 ? 4 : 2) * (size_t)*stride;
}

template <int MODE>
static void launch_long_mode(const SwArgs &a, uint32_t n_tasks, uint32_t pair_base, int32_t *work, uint32_t stride, hipStream_t s) {
    constexpr int BW = ((long_rows_per_lane(MODE) / 4) | 1);
    const size_t lds = (size_t)2 * SW_NLET * LONG_G * BW * 4;
    static PerDeviceOnce once;
    if (lds > 64 * 1024)
        once([&] { (void)hipFuncSetAttribute((const void *)sw_long_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL(sw_long_kernel<MODE>, dim3(n_tasks), dim3(LONG_NW * 64), lds, s, a, pair_base, work, stride);
}

void launch_sw_long(int mode, const SwArgs &a, uint32_t n_tasks, uint32_t pair_base, int32_t *work, uint32_t stride, hipStream_t s) {
    if (n_tasks == 0) return;
    if (mode == 0) launch_long_mode<0>(a, n_tasks, pair_base, work, stride, s);
    else if (mode == 1) launch_long_mode<1>(a, n_tasks, pair_base, work, stride, s);
    else if (mode == 2) launch_long_mode<2>(a, n_tasks, pair_base, work, stride, s);
    else launch_long_mode<3>(a, n_tasks, pair_base, work, stride, s);
}

}  // namespace uc
