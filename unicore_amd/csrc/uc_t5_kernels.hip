// uc_t5_kernels.hip — the ProstT5 AA -> 3Di encoder (BASELINE configs[4]) as hand-written gfx950
// kernels: what `foldseek createdb --prostt5-model ... --gpu 1` runs for /root/reference/src/modules/createdb.rs:157-166.
// T5 encoder (pre-norm blocks: RMSNorm -> self-attention with relative-position bias -> residual; RMSNorm -> ReLU FFN ->
// residual; final RMSNorm) + the two-layer 3Di CNN head of ProstT5 (conv 1024->32 k=7, ReLU, conv 32->20 k=7, argmax).
//
// This is the one dense-contraction stage of the pipeline, so it runs on the matrix cores:
//   * every linear layer is one f16 MFMA GEMM (v_mfma_f32_16x16x32_f16, fp32 accumulate), 128 x 128 x 64 tiles staged
//     through LDS, with the epilogue fused: f16 store (+ReLU) or fp32 accumulation into the residual stream;
//   * attention is a flash-style kernel: S = Q K^T and O = P V on MFMA, online softmax in registers, the T5 bias added
//     from a per-head table indexed by (key - query), K / V tiles shared by the 4 waves of a workgroup through LDS;
//   * the residual stream stays fp32 (T5's known f16 overflow is in that accumulation), GEMM operands are f16 as in the
//     GGUF file (prostt5-f16.gguf).
// Fragment layouts (checked on the hardware by tools/ubench/mfma_layout.hip):
//   A (16 x 32): lane l holds A[l & 15][(l >> 4) * 8 + 0..7];  B (32 x 16): lane l holds B[(l >> 4) * 8 + 0..7][l & 15];
//   C (16 x 16): lane l, register r holds C[(l >> 4) * 4 + r][l & 15].
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "uc_t5.h"

namespace uc {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------- GEMM
// out[M, N] (+)= A[M, K] . W[N, K]^T      A, W f16 row-major (K contiguous: the layout of a torch Linear weight)
// EPI 0: f16 store, 1: ReLU + f16 store, 2: fp32 accumulate into out (the residual stream)
//
// 128 x 128 x 64 tiles, 4 waves x (4 x 4 MFMA tiles of 16 x 16 x 32).  Operands go from global memory STRAIGHT into LDS
// (global_load_lds, 16 bytes per lane: no staging registers, no ds_write), two stages: the loads of K-tile kt + 1 are in
// flight while tile kt is multiplied, ONE barrier per K-step.  A wave's load instruction fills 1 KB = 8 rows of 128 bytes;
// the image is linear (the DMA writes lane-linear) but XOR-swizzled through the GLOBAL address each lane picks: slot c of
// row r holds 16-byte chunk c ^ (r & 7), so the 16 rows of an MFMA fragment read (same chunk, consecutive rows) spread over
// all banks (2-way instead of 16-way conflicts on the 128-byte row stride).
constexpr int GBM = 128, GBN = 128, GBK = 64, GXM = 8;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int EPI>
__global__ void __launch_bounds__(256) t5_gemm_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, void *__restrict__ out,
                                                      int M, int N, int K) {
    __shared__ __attribute__((aligned(1024))) _Float16 sm[2][2][GBM * GBK];      // [stage][A | B][row * 64 + slot * 8]: 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own 4 MB L2: XCD x owns
    // the row tiles x, x + 8, ... and walks them in groups of GXM rows x all column tiles, rows fastest - the ~70 workgroups
    // an XCD runs at a time then share GXM A tiles and a handful of W tiles through ITS L2 instead of every XCD streaming
    // every A row tile (plain row-major order with 8 column tiles put column tile x on XCD x: A was read 8 times over)
    int m0, n0;
    {
        const int nn = (N + GBN - 1) / GBN, nm = (M + GBM - 1) / GBM;
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int g = k / (GXM * nn), r = k % (GXM * nn);
        const int ml = g * GXM + r % GXM, mt = xcd + 8 * ml;
        if (mt >= nm) return;
        m0 = mt * GBM;
        n0 = (r / GXM) * GBN;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's source rows / chunks of the 4 + 4 load instructions of a K-tile (instruction i of wave w fills rows
    // (4 w + i) * 8 .. + 7; lane l -> row + l / 8, slot l % 8, which receives chunk slot ^ (row & 7))
    const _Float16 *ga[4], *gb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3), kc = ((lane & 7) ^ (row & 7)) * 8;
        ga[i] = A + (size_t)min(m0 + row, M - 1) * K + kc;              // rows beyond M are never stored: any valid address will do
        gb[i] = W + (size_t)min(n0 + row, N - 1) * K + kc;
    }
    auto issue = [&](int kt, int st) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __builtin_amdgcn_global_load_lds((gbl_void *)(ga[i] + (size_t)kt * GBK), (lds_void *)(&sm[st][0][(wave * 4 + i) * 8 * GBK]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(gb[i] + (size_t)kt * GBK), (lds_void *)(&sm[st][1][(wave * 4 + i) * 8 * GBK]), 16, 0, 0);
        }
    };
    const int nk = K / GBK;
    issue(0, 0);
    for (int kt = 0; kt < nk; kt++) {
        const int st = kt & 1;
        __syncthreads();                             // tile kt has landed for every wave (the compiler drains vmcnt here) and everybody is done with tile kt - 1
        if (kt + 1 < nk) issue(kt + 1, st ^ 1);
        const _Float16 *sA = sm[st][0], *sB = sm[st][1];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            half8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = wm * 64 + i * 16 + (lane & 15);
                af[i] = *(const half8 *)(sA + r * GBK + (((ks * 4 + (lane >> 4)) ^ (r & 7)) * 8));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r = wn * 64 + j * 16 + (lane & 15);
                bf[j] = *(const half8 *)(sB + r * GBK + (((ks * 4 + (lane >> 4)) ^ (r & 7)) * 8));
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    // the tiles are computed TRANSPOSED (W fragment as the A operand): a lane holds out[row l & 15][4 consecutive columns
    // (l >> 4) * 4 ..], so the epilogue moves 8 / 16 bytes per access instead of 2 / 4
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = m0 + wm * 64 + i * 16 + (lane & 15);
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (col >= N) continue;                                  // (N is a multiple of 4)
            f32x4 v = acc[i][j];
            if (EPI == 2) {
                f32x4 *o = (f32x4 *)((float *)out + (size_t)row * N + col);
                *o = *o + v;
            } else {
                if (EPI == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                *(h4 *)((_Float16 *)out + (size_t)row * N + col) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
    }
}

// ---- the large-tile variant: 256 x 256 x 64, 8 waves x (128 x 64) -------------------------------------------------------
// A wave tile of 128 x 64 reads 24 LDS fragments per 64 MFMAs (0.375 per MFMA instead of the 0.5 of the 128^2 kernel's 64 x 64),
// and the 256 x 256 workgroup tile halves the global -> LDS traffic per FLOP.  Same staging (global_load_lds into a
// double-buffered XOR-swizzled tile, one barrier per K-step), 128 KB of LDS, one workgroup of 8 waves per CU, accumulators 128
// VGPRs.  Measured (r3d): whole encoder 746 -> 835 TFLOP/s, L2 hit rate 79 % (rocprofv3 TCC_HIT / TCC_MISS).  (The r3 reading
// "the LDS port is saturated" assumed 128 B / clock; ds_read_b128 moves 256 B / clock on gfx950 and the fragment rows are
// conflict-free in its 16-lane groups - the decomposition that replaced it stands at t5_gemm256x_kernel.)  Kept as the
// fallback where the two-phase kernel's 32-bit staging offsets do not reach, and as the baseline of tools/t5_gemm_ab.py.
constexpr int HBM_ = 256, HBN_ = 256;

template <int EPI>
__global__ void __launch_bounds__(512) t5_gemm256_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, void *__restrict__ out,
                                                         int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) _Float16 smd[];                // [stage][A | B][256 rows * 64]: 128 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    int m0, n0;
    {
        const int nn = (N + HBN_ - 1) / HBN_, nm = (M + HBM_ - 1) / HBM_;
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int g = k / (GXM * nn), r = k % (GXM * nn);
        const int ml = g * GXM + r % GXM, mt = xcd + 8 * ml;
        if (mt >= nm) return;
        m0 = mt * HBM_;
        n0 = (r / GXM) * HBN_;
    }
    auto tile = [&](int st, int op) -> _Float16 * { return smd + (size_t)(st * 2 + op) * (HBM_ * GBK); };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const _Float16 *ga[4], *gb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3), kc = ((lane & 7) ^ (row & 7)) * 8;
        ga[i] = A + (size_t)min(m0 + row, M - 1) * K + kc;
        gb[i] = W + (size_t)min(n0 + row, N - 1) * K + kc;
    }
    auto issue = [&](int kt, int st) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __builtin_amdgcn_global_load_lds((gbl_void *)(ga[i] + (size_t)kt * GBK), (lds_void *)(tile(st, 0) + (wave * 4 + i) * 8 * GBK), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(gb[i] + (size_t)kt * GBK), (lds_void *)(tile(st, 1) + (wave * 4 + i) * 8 * GBK), 16, 0, 0);
        }
    };
    const int nk = K / GBK;
    issue(0, 0);
    for (int kt = 0; kt < nk; kt++) {
        const int st = kt & 1;
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, st ^ 1);
        const _Float16 *sA = tile(st, 0), *sB = tile(st, 1);
        // fragments of both K-halves in registers: the second half's LDS reads are interleaved with the first half's MFMAs
        // (one ds_read_b128 per 2-3 MFMAs), so only the first 12 reads of a K-step are exposed
        half8 af0[8], bf0[4], af1[8], bf1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = wn * 64 + j * 16 + (lane & 15);
            bf0[j] = *(const half8 *)(sB + r * GBK + (((lane >> 4) ^ (r & 7)) * 8));
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int r = wm * 128 + i * 16 + (lane & 15);
            af0[i] = *(const half8 *)(sA + r * GBK + (((lane >> 4) ^ (r & 7)) * 8));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = wn * 64 + j * 16 + (lane & 15);
            bf1[j] = *(const half8 *)(sB + r * GBK + (((4 + (lane >> 4)) ^ (r & 7)) * 8));
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int r = wm * 128 + i * 16 + (lane & 15);
            af1[i] = *(const half8 *)(sA + r * GBK + (((4 + (lane >> 4)) ^ (r & 7)) * 8));
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf0[j], af0[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf1[j], af1[i], acc[i][j], 0, 0, 0);
        // schedule: 12 DS reads (first half), then 12 x {1 DS read, 2 MFMA} + 8 MFMA, then the second half's 32 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int q = 0; q < 12; q++) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 40, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = m0 + wm * 128 + i * 16 + (lane & 15);
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (col >= N) continue;
            f32x4 v = acc[i][j];
            if (EPI == 2) {
                f32x4 *o = (f32x4 *)((float *)out + (size_t)row * N + col);
                *o = *o + v;
            } else {
                if (EPI == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                *(h4 *)((_Float16 *)out + (size_t)row * N + col) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
    }
}

// ---- the two-phase persistent variant of the 256 x 256 tile (r4) ------------------------------------------------------------------------------
// Same tile, same 8 waves of 128 x 64 and the same bytes through LDS as t5_gemm256_kernel, but a K-step is no longer "wait for the whole tile, barrier,
// read, multiply".  A K-tile is staged as FOUR half-tiles ("pieces") of 16 KB - A-half h = the rows of sub-block h (64 rows) of BOTH row-waves, B-half h =
// the columns of sub-block h (32 columns) of all four column-waves - and worked off in TWO phases of 32 MFMAs:
//   X: stage A-half 1 of step kt + 1 | read A-half 0 and both B halves (16 fragments) | wave-tile rows 0 .. 63
//   Y: stage A-half 0, B-half 0, B-half 1 of step kt + 2 (the pieces X has just freed, in the buffer step kt is still read from) | read A-half 1 (8 fragments) | rows 64 .. 127
// B fragments stay in registers for the whole K-step.  A phase is {stage; LDS reads; counted s_waitcnt vmcnt(8) (four pieces may still be in flight: what the
// NEXT phase reads has landed) + lgkmcnt(0); barrier; 32 MFMAs at raised priority; barrier}, raw s_barrier instead of __syncthreads (whose fence would drain
// the DMA queue), loads in flight ACROSS barriers.  The two row-waves of a SIMD run ONE BARRIER APART (the second executes one extra barrier up front, the first one
// extra at the end), so on every SIMD one wave multiplies while the other reads and stages: the matrix pipe and the LDS port work side by side.  Hazards: a piece
// is re-staged in the phase after its last read, every read is retired (lgkmcnt(0)) before the barrier that ends its interval - both wave groups included (the
// later group reads and stages one interval later); a wait in phase q is followed by both groups' barriers before anybody reads in phase q + 1.
// PERSISTENT: one workgroup per CU (128 KB of LDS) walks its tiles; the K-steps of consecutive tiles form ONE stream - the last steps of a tile stage the first of
// the next (the pieces' staging offsets move on one at a time), so only a workgroup's first tile pays the load latency, and a tile's stores drain under the next tile.
// Where the time goes (tools/ubench/gemm_supply.hip = this kernel with parts switched off; TFLOP/s-equivalent at 65,536 tokens, K = 1024 / 16384;
// profiles/r03_gemm_supply.log): MFMA loop alone 1.8 / 2.05 P (the clock under MFMA load is ~2.0 GHz: 2.05 P IS the matrix pipe), + epilogue stores 1.46 / 2.05,
// + LDS fragment reads 1.26 / 1.76, + the global -> LDS stream 0.9-1.1 / 1.34.  The stream alone runs at 12-14 TB/s (20-23 B / clock / CU) and 1.3-1.8x faster
// when every load hits L2; more bytes in flight do not speed it up (bandwidth-, not latency-bound).  The K order per output element is unchanged: bit-identical
// to the other GEMM kernels.
template <int EPI>
__global__ void __launch_bounds__(512) t5_gemm256x_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, void *__restrict__ out,
                                                          int M, int N, int K, int gxm, int slots) {
    extern __shared__ __attribute__((aligned(1024))) _Float16 smx[];                // [buffer][A0 | A1 | B0 | B1][128 rows * 64]: 128 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    const int nn = N / HBN_, nm = (M + HBM_ - 1) / HBM_, xcd = blockIdx.x & 7;
    const int n_local = ((nm + 7) / 8 + gxm - 1) / gxm * gxm * nn;
    auto decode = [&](int k, int &tm0, int &tn0) -> bool {
        const int g = k / (gxm * nn), r = k % (gxm * nn);
        const int mt = xcd + 8 * (g * gxm + r % gxm);
        tm0 = mt * HBM_;
        tn0 = (r / gxm) * HBN_;
        return mt < nm;
    };
    auto seek = [&](int k, int &tm0, int &tn0) -> int {
        while (k < n_local && !decode(k, tm0, tn0)) k += slots;
        return k;
    };
    int m0 = 0, n0 = 0, kcur = seek((int)(blockIdx.x >> 3), m0, n0);
    if (kcur >= n_local) return;
    constexpr int HT = 128 * GBK;
    auto half = [&](int buf, int which) -> _Float16 * { return smx + (size_t)(buf * 4 + which) * HT; };   // which: 0 A0, 1 A1, 2 B0, 3 B1
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t oa[2][2], ob[2][2];
    auto set_a = [&](int h, int tm0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = (wave * 2 + i) * 8 + (lane >> 3), kc = ((lane & 7) ^ (r & 7)) * 8;
            oa[h][i] = (uint32_t)(((size_t)min(tm0 + (r >> 6) * 128 + h * 64 + (r & 63), M - 1) * K + kc) * 2);
        }
    };
    auto set_b = [&](int h, int tn0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = (wave * 2 + i) * 8 + (lane >> 3), kc = ((lane & 7) ^ (r & 7)) * 8;
            ob[h][i] = (uint32_t)(((size_t)(tn0 + (r >> 5) * 64 + h * 32 + (r & 31)) * K + kc) * 2);
        }
    };
    set_a(0, m0); set_a(1, m0); set_b(0, n0); set_b(1, n0);
    auto stage = [&](int kt, int buf, int which) {
        const char *base = (which < 2 ? (const char *)A : (const char *)W) + (size_t)kt * (GBK * 2);
        _Float16 *dst = half(buf, which);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t o = which < 2 ? oa[which][i] : ob[which - 2][i];
            __builtin_amdgcn_global_load_lds((gbl_void *)(base + o), (lds_void *)(dst + (wave * 2 + i) * 8 * GBK), 16, 0, 0);
        }
    };
    auto frag = [&](const _Float16 *s, int r0, int ks) -> half8 {
        const int r = r0 + (lane & 15);
        return *(const half8 *)(s + r * GBK + (((ks * 4 + (lane >> 4)) ^ (r & 7)) * 8));
    };
    const int nk = K / GBK;                                                         // >= 2 (the launcher checks)
    // prologue: step 0 and, of step 1, what phase Y of "step -1" would have staged
    stage(0, 0, 0); stage(0, 0, 2); stage(0, 0, 3); stage(0, 0, 1);
    stage(1, 1, 0); stage(1, 1, 2); stage(1, 1, 3);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                                      // the second row-wave of every SIMD runs one barrier behind

    half8 fa[4][2], fb[4][2];
    int buf = 0;
    for (;;) {                                                                      // this workgroup's tiles
    int m0n = 0, n0n = 0;
    const int knext = seek(kcur + slots, m0n, n0n);
    const bool has_next = knext < n_local;
    for (int kt = 0; kt < nk; kt++, buf ^= 1) {
        // the staging offsets move on to the next tile piece by piece: phase Y stages two steps ahead, phase X one
        if (has_next && kt == nk - 2) { set_a(0, m0n); set_b(0, n0n); set_b(1, n0n); }
        if (has_next && kt == nk - 1) set_a(1, m0n);
        const bool more1 = kt + 1 < nk || has_next, more2 = kt + 2 < nk || has_next;
        const int kt1 = kt + 1 < nk ? kt + 1 : 0, kt2 = kt + 2 < nk ? kt + 2 : kt + 2 - nk;
        const bool tail = !has_next && kt + 2 >= nk;                               // the stream ends: nothing is staged behind what is waited for
        const _Float16 *sA0 = half(buf, 0) + wr * 64 * GBK, *sA1 = half(buf, 1) + wr * 64 * GBK;
        const _Float16 *sB0 = half(buf, 2) + wc * 32 * GBK, *sB1 = half(buf, 3) + wc * 32 * GBK;
        // ---- phase X
        if (more1) stage(kt1, buf ^ 1, 1);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) { fb[j][ks] = frag(sB0, j * 16, ks); fb[2 + j][ks] = frag(sB1, j * 16, ks); }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) fa[i][ks] = frag(sA0, i * 16, ks);
        if (tail) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][ks], fa[i][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase Y
        if (more2) { stage(kt2, buf, 0); stage(kt2, buf, 2); stage(kt2, buf, 3); }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) fa[i][ks] = frag(sA1, i * 16, ks);
        if (tail) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][ks], fa[i][ks], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    // epilogue (no barrier inside: the wave groups stay one barrier apart; one group stores while the other multiplies).  A lane holds out[row l & 15][4
    // consecutive columns (l >> 4) * 4 ..] of each 16 x 16 tile: stored as they are, the f16 results leave as 8 bytes per lane = 32-byte pieces of 16 rows per
    // instruction, and such stores cost the K = 1024 layers a quarter of their time (gemm_supply: 0.95 -> 1.17-1.22 P with 64- / 128-byte row pieces).  So two
    // neighbouring tiles are interleaved first: v_permlane16_swap hands lane group 1's quarter of tile j to group 0 and group 0's quarter of tile j + 1 to
    // group 1 (likewise 3 <-> 2), after which every lane owns 8 consecutive columns = 16 bytes, and an instruction writes 64 contiguous bytes of 16 rows.
    if (EPI == 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = m0 + wr * 128 + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int col = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
                const f32x4 v = acc[i][j];
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row >= M) continue;
                f32x4 *o = (f32x4 *)((float *)out + (size_t)row * N + col);
                *o = *o + v;
            }
        }
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const int g = lane >> 4;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = m0 + wr * 128 + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                f32x4 v0 = acc[i][j], v1 = acc[i][j + 1];
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[i][j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (EPI == 1) {
#pragma unroll
                    for (int q = 0; q < 4; q++) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
                }
                const uint32_t ax = __builtin_bit_cast(uint32_t, h2{(_Float16)v0[0], (_Float16)v0[1]}), ay = __builtin_bit_cast(uint32_t, h2{(_Float16)v0[2], (_Float16)v0[3]});
                const uint32_t bx = __builtin_bit_cast(uint32_t, h2{(_Float16)v1[0], (_Float16)v1[1]}), by = __builtin_bit_cast(uint32_t, h2{(_Float16)v1[2], (_Float16)v1[3]});
                const auto sx = __builtin_amdgcn_permlane16_swap(ax, bx, false, false), sy = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
                // group g now holds tile j + (g & 1), columns (g >> 1) * 8 .. + 7: first result = the lower four, second = the upper four
                const int col = n0 + wc * 64 + (j + (g & 1)) * 16 + (g >> 1) * 8;
                if (row < M) *(u32x4 *)((_Float16 *)out + (size_t)row * N + col) = u32x4{sx[0], sy[0], sx[1], sy[1]};
            }
        }
    }
    if (!has_next) break;
    m0 = m0n; n0 = n0n; kcur = knext;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
}

template <int EPI>
static void t5_gemm256x_launch(const void *A, const void *W, void *out, int M, int N, int K, hipStream_t s) {
    constexpr int LDS = 2 * 4 * 128 * GBK * 2;
    static bool once[64] = {};
    static int cus[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!once[dev]) {
        (void)hipFuncSetAttribute((const void *)t5_gemm256x_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipDeviceProp_t pr;
        cus[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount >= 8 ? pr.multiProcessorCount : 256;
        once[dev] = true;
    }
    const int gxm = 2;     // M-tiles per XCD group (swept in r03: profiles/r03_t5_gxm*.log)
    const int nn = N / HBN_, nm = (M + HBM_ - 1) / HBM_, n_local = ((nm + 7) / 8 + gxm - 1) / gxm * gxm * nn;
    const int slots = std::min(n_local, std::max(1, cus[dev] / 8));
    hipLaunchKernelGGL(t5_gemm256x_kernel<EPI>, dim3((unsigned)(8 * slots)), dim3(512), LDS, s, (const _Float16 *)A, (const _Float16 *)W, out, M, N, K, gxm, slots);
}

template <int EPI>
static void t5_gemm256_launch(const void *A, const void *W, void *out, int M, int N, int K, hipStream_t s) {
    constexpr int LDS = 2 * 2 * HBM_ * GBK * 2;
    static bool once[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !once[dev]) {
        (void)hipFuncSetAttribute((const void *)t5_gemm256_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        once[dev] = true;
    }
    const int nn = (N + HBN_ - 1) / HBN_, nm = (M + HBM_ - 1) / HBM_, per_xcd = ((nm + 7) / 8 + GXM - 1) / GXM * GXM;
    hipLaunchKernelGGL(t5_gemm256_kernel<EPI>, dim3((unsigned)(8 * per_xcd * nn)), dim3(512), LDS, s, (const _Float16 *)A, (const _Float16 *)W, out, M, N, K);
}

void t5_gemm(int epi, const void *A, const void *W, void *out, int M, int N, int K, hipStream_t s) {
    if (M <= 0) return;
    // UC_T5_GEMM256: 0 = 128 x 128 tile only, 1 = 256 x 256 tile, 8 waves of 128 x 64, one barrier per K-step (r3), 2 (default) = the same tile worked off in
    // two phases per K-step by persistent workgroups (t5_gemm256x_kernel).  All sum K in the same order: bit-identical results (tools/t5_gemm_ab.py).
    // Measured and NOT kept in the library (r4; numbers in profiles/r03_t5_gemm_ab*.json, profiles/r03_gemm_supply.log): a 256 x 256 tile with FOUR waves of
    // 128 x 128 and AGPR accumulators (tools/experiments/uc_t5_gemm4w.hip: 693 TFLOP/s for the 24-block encoder against 739 with the 128 x 128 tile - one wave
    // per SIMD leaves nobody to cover its barrier and wait stalls); four phases of 16 MFMAs per K-step (+3 % over the one-barrier kernel, +5.5 % persistent;
    // superseded by the two-phase kernel, +8.5 %); a ten-slot ring of half-tiles over all 160 KB of LDS (seven half-tiles in flight: 5 % slower); a start skew
    // between the workgroups of an XCD (no effect); a wait that names the epilogue's stores so that loads issued before them are not held up (no effect).
    static const int big = getenv("UC_T5_GEMM256") ? atoi(getenv("UC_T5_GEMM256")) : 2;
    if (big >= 2 && M >= 2048 && N % HBN_ == 0 && K % GBK == 0 && K >= 2 * GBK && (size_t)M * K * 2 < (1ull << 32) && (size_t)N * K * 2 < (1ull << 32)) {
        if (epi == 0) t5_gemm256x_launch<0>(A, W, out, M, N, K, s);
        else if (epi == 1) t5_gemm256x_launch<1>(A, W, out, M, N, K, s);
        else t5_gemm256x_launch<2>(A, W, out, M, N, K, s);
        return;
    }
    if (big && M >= 2048 && N % HBN_ == 0) {     // large batches: the 256 x 256 tile (small ones would leave most CUs without a tile)
        if (epi == 0) t5_gemm256_launch<0>(A, W, out, M, N, K, s);
        else if (epi == 1) t5_gemm256_launch<1>(A, W, out, M, N, K, s);
        else t5_gemm256_launch<2>(A, W, out, M, N, K, s);
        return;
    }
    const int nn = (N + GBN - 1) / GBN, nm = (M + GBM - 1) / GBM, per_xcd = ((nm + 7) / 8 + GXM - 1) / GXM * GXM;
    const dim3 grid((unsigned)(8 * per_xcd * nn));
    if (epi == 0) hipLaunchKernelGGL(t5_gemm_kernel<0>, grid, dim3(256), 0, s, (const _Float16 *)A, (const _Float16 *)W, out, M, N, K);
    else if (epi == 1) hipLaunchKernelGGL(t5_gemm_kernel<1>, grid, dim3(256), 0, s, (const _Float16 *)A, (const _Float16 *)W, out, M, N, K);
    else hipLaunchKernelGGL(t5_gemm_kernel<2>, grid, dim3(256), 0, s, (const _Float16 *)A, (const _Float16 *)W, out, M, N, K);
}

// ---------------------------------------------------------------------------------------------- embedding / RMSNorm
__global__ void __launch_bounds__(256) t5_embed_kernel(const int32_t *__restrict__ tok, const _Float16 *__restrict__ emb, float *__restrict__ hidden,
                                                       int T, int D, int vocab) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)T * D; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i / D), d = (int)(i % D);
        int id = tok[t];
        id = id < 0 || id >= vocab ? 0 : id;
        hidden[i] = (float)emb[(size_t)id * D + d];
    }
}
void t5_embed(const int32_t *tok, const void *emb, float *hidden, int T, int D, int vocab, hipStream_t s) {
    if (T <= 0) return;
    hipLaunchKernelGGL(t5_embed_kernel, dim3((unsigned)std::min<size_t>(((size_t)T * D + 255) / 256, 65535)), dim3(256), 0, s, tok, (const _Float16 *)emb, hidden, T, D, vocab);
}

// T5LayerNorm: y = x * rsqrt(mean(x^2) + eps) * w   (no mean subtraction, no bias); one wave per token
__global__ void __launch_bounds__(256) t5_rmsnorm_kernel(const float *__restrict__ x, const float *__restrict__ w, _Float16 *__restrict__ y, int T, int D, float eps) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float *xr = x + (size_t)t * D;
    float ss = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 v = *(const float4 *)(xr + d);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float sc = rsqrtf(ss / (float)D + eps);
    _Float16 *yr = y + (size_t)t * D;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 v = *(const float4 *)(xr + d), g = *(const float4 *)(w + d);
        yr[d] = (_Float16)(v.x * sc * g.x); yr[d + 1] = (_Float16)(v.y * sc * g.y);
        yr[d + 2] = (_Float16)(v.z * sc * g.z); yr[d + 3] = (_Float16)(v.w * sc * g.w);
    }
}
void t5_rmsnorm(const float *x, const float *w, void *y, int T, int D, float eps, hipStream_t s) {
    if (T <= 0) return;
    hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3((T + 3) / 4), dim3(256), 0, s, x, w, (_Float16 *)y, T, D, eps);
}

// ---------------------------------------------------------------------------------------------- attention
// qkv: [T, 3 * H * 128] f16 (q | k | v), out: [T, H * 128] f16.  bias: [H][2 * bias_span - 1] fp32, entry (key - query) +
// bias_span - 1.  T5 applies NO 1/sqrt(d) scaling.  d_kv = 128.
//
// Both contractions are computed TRANSPOSED so that nothing has to be re-laid-out between them and no LDS is needed:
//   S^T = K . Q^T   (A = K rows: 16 keys x 32 dims, B = Q^T: this lane's query l & 15, 8 contiguous dims)
//         -> lane holds S[query l & 15][keys (l >> 4) * 4 + r] of each 16-key sub-tile: a query's softmax statistics live in
//            the 4 lanes {q, q + 16, q + 32, q + 48}: in-lane reduction over 8 values + two xor-shuffles (16, 32);
//   O^T = V^T . P^T (A = V^T rows: 16 dims x 32 keys, B = P^T: the very registers the softmax left, as f16)
//         -> lane holds O[query l & 15][dims (l >> 4) * 4 + r]: the same query as its statistics, so the online-softmax
//            rescale is a plain per-lane multiply and the result leaves as 8-byte stores.
// The 32 keys of a block enter the second contraction in the order the first one produced them (k = g * 8 + e <-> key
// g * 4 + e for e < 4, 16 + g * 4 + e - 4 otherwise); a sum does not care.
// V is consumed ROW-MAJOR, as the q|k|v GEMM wrote it: the A fragment of V^T (16 dims x those 8 keys per lane) comes out of two
// ds_read_b64_tr_b16 - gfx950's transposing LDS read: within a 16-lane group, lane i receives element i % 4 of what lanes i / 4, 4 + i / 4,
// 8 + i / 4, 12 + i / 4 address (checked on the hardware), i.e. column i of the 4 keys x 16 dims block whose row j the lanes 4 j .. 4 j + 3
// point at.  (Until r4 a separate pass wrote a transposed, padded copy V^T[H * 128][tokens] per layer: 3.6 % of the encoder, at HBM speed.)
constexpr int ADK = 128, AQW = 32;            // head dim, keys per block, queries per wave (2 tiles of 16)
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4 lds_fp16x4;

// One workgroup = 128 queries of one (sequence, head): 4 waves x 32 queries.  The K block and the V block (64 keys x 128 dims
// each) are loaded ONCE per workgroup with coalesced 16-byte loads (the next block travels in registers
// while this one is multiplied) and shared through LDS: per-wave fragment loads straight from L2 re-read every block once
// per 32 queries and were bandwidth / address-bound (77 TFLOP/s; profiles/round2/r3_t5).
constexpr int AKB = 64;                                  // keys per block
constexpr int SK_LD = ADK + 8, SV_LD = ADK + 16;         // LDS row strides in halves: K 272 B; V 288 B = 32 (mod 256): the 8 keys x 32 bytes a half-wave's
                                                         // transposing read touches fall into 8 different 32-byte bank groups

__global__ void __launch_bounds__(256) t5_attention_kernel(const _Float16 *__restrict__ qkv, const T5AttnTile *__restrict__ tiles, const float *__restrict__ bias, int bias_span, int H,
                                                           _Float16 *__restrict__ out) {
    // two buffers: while a block is multiplied, the next one's registers are written into the other buffer as soon as its loads have landed - ONE barrier per block
    __shared__ __attribute__((aligned(16))) _Float16 sK2[2][AKB * SK_LD];
    __shared__ __attribute__((aligned(16))) _Float16 sV2[2][AKB * SV_LD];
    __shared__ float sB2[2][192];                         // the bias entries this workgroup needs for one key block: (key - query) = k0 - q0 - 127 .. k0 - q0 + 63
    const T5AttnTile tl = tiles[blockIdx.x];
    const int h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int L = tl.len, q0 = tl.q0 + w * AQW;
    const size_t ld = (size_t)3 * H * ADK;
    const _Float16 *qb = qkv + (size_t)tl.tok0 * ld + (size_t)h * ADK, *kb = qb + (size_t)H * ADK, *vb = kb + (size_t)H * ADK;
    const float *bh = bias + (size_t)h * (2 * bias_span - 1) + (bias_span - 1);
    half8 qf[2][4];                                      // B fragments of Q^T: query q0 + qt * 16 + c, dims kk * 32 + g * 8 ..
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        const int qi = q0 + qt * 16 + c;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) qf[qt][kk] = qi < L ? *(const half8 *)(qb + (size_t)qi * ld + kk * 32 + g * 8) : half8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    f32x4 o[2][8];
#pragma unroll
    for (int qt = 0; qt < 2; qt++)
#pragma unroll
        for (int d = 0; d < 8; d++) o[qt][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrow[2] = {-INFINITY, -INFINITY}, lrow[2] = {0.f, 0.f};
    // staging: K block and V block = 64 rows x 16 chunks of 16 B each: 4 + 4 chunks per thread.  All eight loads are unconditional (key rows
    // clamped: keys >= L are masked in the softmax and their P is exactly 0) and issued back to back - as lambdas over register arrays under
    // `key < L` they were compiled to branchy, serialized loads spilled through scratch memory
    const int skey = tid >> 4, sdc = (tid & 15) * 8;                 // thread -> (key row, dim chunk); + 16 keys per i
    // the next block's K / V chunks travel in registers while this block is multiplied (T14: issue early, write late)
    uint4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
    float rb = 0.f;                                      // thread t < 192: bias of (key - query) = k0 - tl.q0 - 127 + t.  (Read per lane from global memory
    // inside the softmax, 8 x 16 bytes per block, the bias loads sat BEHIND the next block's K / V^T prefetch in the in-order vmcnt queue: every block waited for
    // its successor's loads - 311 TFLOP/s.)
    auto fetch = [&](int k0) {
        if (tid < 192) rb = bh[max(-(bias_span - 1), min(bias_span - 1, k0 - tl.q0 - 127 + tid))];
        rk0 = *(const uint4 *)(kb + (size_t)min(k0 + skey, L - 1) * ld + sdc);
        rk1 = *(const uint4 *)(kb + (size_t)min(k0 + skey + 16, L - 1) * ld + sdc);
        rk2 = *(const uint4 *)(kb + (size_t)min(k0 + skey + 32, L - 1) * ld + sdc);
        rk3 = *(const uint4 *)(kb + (size_t)min(k0 + skey + 48, L - 1) * ld + sdc);
        rv0 = *(const uint4 *)(vb + (size_t)min(k0 + skey, L - 1) * ld + sdc);
        rv1 = *(const uint4 *)(vb + (size_t)min(k0 + skey + 16, L - 1) * ld + sdc);
        rv2 = *(const uint4 *)(vb + (size_t)min(k0 + skey + 32, L - 1) * ld + sdc);
        rv3 = *(const uint4 *)(vb + (size_t)min(k0 + skey + 48, L - 1) * ld + sdc);
    };
    auto stash = [&](int b) {                            // the fetched registers -> buffer b
        _Float16 *dK = sK2[b], *dV = sV2[b];
        *(uint4 *)(dK + skey * SK_LD + sdc) = rk0;
        *(uint4 *)(dK + (skey + 16) * SK_LD + sdc) = rk1;
        *(uint4 *)(dK + (skey + 32) * SK_LD + sdc) = rk2;
        *(uint4 *)(dK + (skey + 48) * SK_LD + sdc) = rk3;
        *(uint4 *)(dV + skey * SV_LD + sdc) = rv0;
        *(uint4 *)(dV + (skey + 16) * SV_LD + sdc) = rv1;
        *(uint4 *)(dV + (skey + 32) * SV_LD + sdc) = rv2;
        *(uint4 *)(dV + (skey + 48) * SV_LD + sdc) = rv3;
        if (tid < 192) sB2[b][tid] = rb;
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < L; k0 += AKB, cur ^= 1) {
        const _Float16 *sK = sK2[cur], *sV = sV2[cur];
        const float *sB = sB2[cur];
        if (k0 + AKB < L) fetch(k0 + AKB);               // every wave fetches (uniform branch): in flight under the MFMAs below
        if (q0 < L) {                                    // (waves without queries only help with the staging)
            f32x4 sacc[2][4];                            // [query tile][16-key sub-tile]
#pragma unroll
            for (int st = 0; st < 4; st++) {
                sacc[0][st] = f32x4{0.f, 0.f, 0.f, 0.f};
                sacc[1][st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const half8 kf = *(const half8 *)(sK + (st * 16 + c) * SK_LD + kk * 32 + g * 8);
                    sacc[0][st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[0][kk], sacc[0][st], 0, 0, 0);
                    sacc[1][st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[1][kk], sacc[1][st], 0, 0, 0);
                }
            }
            half8 pf[2][2];                              // [query tile][32-key half]
            float alpha[2];
#pragma unroll
            for (int qt = 0; qt < 2; qt++) {
                float p[16], mx = -INFINITY;
                // sB index of (key k0 + st * 16 + g * 4 + r, query tl.q0 + w * 32 + qt * 16 + c): it depends on st - qt only, five distinct groups of 4
                const float *bq = sB + (127 + g * 4 - w * AQW - c);
#pragma unroll
                for (int st = 0; st < 4; st++)
#pragma unroll
                    for (int r = 0; r < 4; r++) p[st * 4 + r] = sacc[qt][st][r] + bq[(st - qt) * 16 + r];     // (unconditional LDS reads: no branches)
                if (k0 + AKB > L) {                      // (uniform) only a sequence's last block has keys to mask
                    asm volatile("" ::: "memory");       // (keeps this a branch: if-converted, the 32 compares + selects ran for every block)
#pragma unroll
                    for (int st = 0; st < 4; st++)
#pragma unroll
                        for (int r = 0; r < 4; r++) p[st * 4 + r] = k0 + st * 16 + g * 4 + r < L ? p[st * 4 + r] : -INFINITY;
                }
#pragma unroll
                for (int e = 0; e < 16; e++) mx = fmaxf(mx, p[e]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(mrow[qt], mx);
                alpha[qt] = __expf(mrow[qt] - mnew);      // first block: exp(-inf) = 0
                constexpr float LOG2E = 1.44269504088896340736f;
                const float mneg = -mnew * LOG2E;
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; e++) { p[e] = __builtin_amdgcn_exp2f(fmaf(p[e], LOG2E, mneg)); sum += p[e]; pf[qt][e >> 3][e & 7] = (_Float16)p[e]; }   // exp(p - mnew): one FMA + v_exp_f32
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                lrow[qt] = lrow[qt] * alpha[qt] + sum;
                mrow[qt] = mnew;
            }
#pragma unroll
            for (int d = 0; d < 8; d++) {
#pragma unroll
                for (int qt = 0; qt < 2; qt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) o[qt][d][r] *= alpha[qt];
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    // A fragment of V^T: dim d * 16 + c, keys in the order of pf: [32 hf + g*4 .. +3 | 32 hf + 16 + g*4 .. +3].  This lane points at
                    // key 32 hf (+ 16) + g * 4 + c / 4, dims d * 16 + (c % 4) * 4 .. + 3 and receives dim d * 16 + c of the group's four keys
                    const _Float16 *vr = sV + (32 * hf + g * 4 + (c >> 2)) * SV_LD + d * 16 + (c & 3) * 4;
                    const half4 v0 = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4 *)vr));
                    const half4 v1 = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4 *)(vr + 16 * SV_LD)));
                    const half8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[0][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[0][hf], o[0][d], 0, 0, 0);
                    o[1][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[1][hf], o[1][d], 0, 0, 0);
                }
            }
        }
        if (k0 + AKB < L) stash(cur ^ 1);                // nobody reads that buffer any more: its last readers passed the previous barrier
        __syncthreads();
    }
    if (q0 >= L) return;
    _Float16 *ob = out + (size_t)tl.tok0 * ((size_t)H * ADK) + (size_t)h * ADK;
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        const int qi = q0 + qt * 16 + c;
        if (qi >= L) continue;
        const float inv = 1.f / lrow[qt];
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const half4 v = {(_Float16)(o[qt][d][0] * inv), (_Float16)(o[qt][d][1] * inv), (_Float16)(o[qt][d][2] * inv), (_Float16)(o[qt][d][3] * inv)};
            *(half4 *)(ob + (size_t)qi * ((size_t)H * ADK) + d * 16 + g * 4) = v;
        }
    }
}
void t5_attention(const void *qkv, const T5AttnTile *tiles, int n_tiles, const float *bias, int bias_span, int H, void *out, hipStream_t s) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(t5_attention_kernel, dim3(n_tiles, H), dim3(256), 0, s, (const _Float16 *)qkv, tiles, bias, bias_span, H, (_Float16 *)out);
}

// ---------------------------------------------------------------------------------------------- 3Di CNN head
// y: [T, ldy] f16 = X . W1r^T with W1r[k * C1 + c][in] = conv1.weight[c][in][k]  (one GEMM for all 7 taps);
// h1[t][c] = relu(b1[c] + sum_k y[t + k - 3][k * C1 + c]).  The convolution runs over the residues and </s> of the
// token's own sequence with zero padding at both ends: ProstT5 slices the <AA2fold> prefix off BEFORE the CNN
// (predict_3Di: residue_embedding[:, 1:]), so the first residue sees zeros on its left, the last one sees </s>.
__global__ void __launch_bounds__(256) t5_conv_h1_kernel(const _Float16 *__restrict__ y, int ldy, const int32_t *__restrict__ seq_of, const int32_t *__restrict__ seq_off,
                                                         const float *__restrict__ b1, float *__restrict__ h1, int T, int C1, int KW, int eos_in_head) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)T * C1; i += (size_t)gridDim.x * 256) {
        // taps reach the residues only; </s> (the last token of the sequence) contributes its hidden state only in the r2 convention —
        // by default it is masked to zero before the head (its OWN position still gets a conv1 output that conv2 sees)
        const int t = (int)(i / C1), c = (int)(i % C1), s = seq_of[t], lo = seq_off[s] + 1, hi = seq_off[s + 1] - (eos_in_head ? 0 : 1);
        float a = b1[c];
        for (int k = 0; k < KW; k++) {
            const int u = t + k - KW / 2;
            if (u >= lo && u < hi) a += (float)y[(size_t)u * ldy + k * C1 + c];
        }
        h1[i] = a > 0.f ? a : 0.f;
    }
}
// logits[t][o] = b2[o] + sum_k sum_c h1[t + k - 3][c] * w2[o][c][k]; codes[t] = argmax_o (first maximum)
__global__ void __launch_bounds__(256) t5_conv2_argmax_kernel(const float *__restrict__ h1, const int32_t *__restrict__ seq_of, const int32_t *__restrict__ seq_off,
                                                              const float *__restrict__ w2, const float *__restrict__ b2, uint8_t *__restrict__ codes,
                                                              float *__restrict__ logits /* nullable: [T, NO] */, int T, int C1, int KW, int NO) {
    for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
        const int s = seq_of[t], lo = seq_off[s] + 1, hi = seq_off[s + 1];
        float best = -INFINITY;
        int arg = 0;
        for (int o = 0; o < NO; o++) {
            float a = b2[o];
            for (int k = 0; k < KW; k++) {
                const int u = t + k - KW / 2;
                if (u < lo || u >= hi) continue;
                const float *hr = h1 + (size_t)u * C1, *wr = w2 + ((size_t)o * C1) * KW + k;
                for (int c = 0; c < C1; c++) a += hr[c] * wr[(size_t)c * KW];
            }
            if (logits) logits[(size_t)t * NO + o] = a;
            if (a > best) { best = a; arg = o; }
        }
        codes[t] = (uint8_t)arg;
    }
}
void t5_cnn_head(const void *y, int ldy, const int32_t *seq_of, const int32_t *seq_off, const float *b1, const float *w2, const float *b2, float *h1,
                 uint8_t *codes, float *logits, int T, int C1, int KW, int NO, int eos_in_head, hipStream_t s) {
    if (T <= 0) return;
    hipLaunchKernelGGL(t5_conv_h1_kernel, dim3((unsigned)std::min<size_t>(((size_t)T * C1 + 255) / 256, 65535)), dim3(256), 0, s, (const _Float16 *)y, ldy, seq_of, seq_off, b1, h1, T, C1, KW, eos_in_head);
    hipLaunchKernelGGL(t5_conv2_argmax_kernel, dim3((T + 255) / 256), dim3(256), 0, s, h1, seq_of, seq_off, w2, b2, codes, logits, T, C1, KW, NO);
}

// f32 -> f16 (weights given as F32 in a GGUF file)
__global__ void __launch_bounds__(256) t5_f32_to_f16_kernel(const float *__restrict__ x, _Float16 *__restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (_Float16)x[i];
}
void t5_f32_to_f16(const float *x, void *y, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(t5_f32_to_f16_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, s, x, (_Float16 *)y, n);
}

}  // namespace uc
