// uc_t5_gemm4w.hip — the ProstT5 encoder's linear layers on a 256 x 256 x 64 tile with FOUR waves of 128 x 128 each (r4).
//
// Why another tile shape: in the 8-wave kernel of uc_t5_kernels.hip (128 x 64 per wave) a K-step moves 192 KB of MFMA fragments
// out of LDS + 64 KB of DMA writes into it = 2048 cycles of the CU's 128 B / cycle LDS port — exactly the MFMA time of the step,
// so neither pipe can be full.  LDS fragment traffic per K-step is (wave rows + wave columns) x 64 x 2 B per wave: a 128 x 128 wave
// tile needs 32 KB per wave, 128 KB per workgroup (+ the same 64 KB of DMA) = 1536 cycles against the same 2048 MFMA cycles.
// The price is the register file: 128 x 128 fp32 accumulators are 256 registers per lane.  gfx950's register file is unified
// (512 per lane at one wave per SIMD): the accumulators live in the 256 AGPRs (MFMA reads and writes them directly), the
// fragments in VGPRs.  This translation unit is therefore compiled WITHOUT -amdgpu-mfma-vgpr-form (Makefile), and with one wave
// per SIMD everything that hides latency has to come from inside the wave: the K-loop is software-pipelined by hand (see the loop).
// Same staging as the 8-wave kernel: global_load_lds (16 B per lane) into a double-buffered, XOR-swizzled (through the GLOBAL
// address) LDS image, one barrier per K-step; same XCD-aware tile order; same fused epilogues; the K order per output element
// is unchanged, so results are bit-identical to the other two GEMM kernels (tests/test_t5.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "uc_t5.h"

namespace uc {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

namespace {
constexpr int QBM = 256, QBN = 256, QBK = 64, QXM = 8;

template <int EPI>
__global__ void __launch_bounds__(256, 1) t5_gemm4w_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, void *__restrict__ out,
                                                           int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) _Float16 smq[];                // [stage][A | B][256 rows * 64]: 128 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int m0, n0;
    {
        const int nn = (N + QBN - 1) / QBN, nm = (M + QBM - 1) / QBM;
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int g = k / (QXM * nn), r = k % (QXM * nn);
        const int ml = g * QXM + r % QXM, mt = xcd + 8 * ml;
        if (mt >= nm) return;
        m0 = mt * QBM;
        n0 = (r / QXM) * QBN;
    }
    auto tile = [&](int st, int op) -> _Float16 * { return smq + (size_t)(st * 2 + op) * (QBM * QBK); };
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 256 rows x 128 B per operand and stage = 32 wave-instructions of 8 rows: 8 per wave and operand
    // per-lane BYTE offsets (32 bits: M x K x 2 < 4 GiB for every layer of the model at the 65,536-token batch limit; the host checks)
    // against the wave-uniform operand bases: 16 VGPRs instead of 16 64-bit pointers — the accumulators leave no room for those
    uint32_t oa[8], ob[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = (wave * 8 + i) * 8 + (lane >> 3), kc = ((lane & 7) ^ (row & 7)) * 8;
        oa[i] = (uint32_t)(((size_t)min(m0 + row, M - 1) * K + kc) * 2);   // rows beyond M are never stored: any valid address will do
        ob[i] = (uint32_t)(((size_t)min(n0 + row, N - 1) * K + kc) * 2);
    }
    auto issue = [&](int kt, int st) {
        const char *Ab = (const char *)A + (size_t)kt * (QBK * 2), *Wb = (const char *)W + (size_t)kt * (QBK * 2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            __builtin_amdgcn_global_load_lds((gbl_void *)(Ab + oa[i]), (lds_void *)(tile(st, 0) + (wave * 8 + i) * 8 * QBK), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(Wb + ob[i]), (lds_void *)(tile(st, 1) + (wave * 8 + i) * 8 * QBK), 16, 0, 0);
        }
    };
    // fragment of 16 rows x 32 k (k-slice ks of the 64-wide tile): rows r0 .. r0 + 15, lane l reads row l & 15, chunk ks * 4 + (l >> 4)
    auto frag = [&](const _Float16 *s, int r0, int ks) -> half8 {
        const int r = r0 + (lane & 15);
        return *(const half8 *)(s + r * QBK + (((ks * 4 + (lane >> 4)) ^ (r & 7)) * 8));
    };
    // Software pipeline across K-steps (one wave per SIMD: nobody else hides this wave's latencies).  Per K-step kt:
    //   16 x {1 read of slice 1, 3 MFMA of slice 0} | 16 MFMA of slice 0, 32 of slice 1 |
    //   wait + barrier: tile kt + 1 has landed everywhere, everybody has read ALL of tile kt |
    //   issue the loads of tile kt + 2 into the stage tile kt just left (a full K-step ahead of the barrier that waits for them) |
    //   16 x {1 read of slice 0 of tile kt + 1, 1 MFMA of slice 1} | 16 MFMA of slice 1
    // so every LDS read and every global load is issued under MFMAs that do not depend on it.
    const int nk = K / QBK;
    issue(0, 0);
    __syncthreads();
    if (nk > 1) issue(1, 1);
    half8 a0[8], b0[8], a1[8], b1[8];
    {
        const _Float16 *sA = tile(0, 0) + (size_t)wm * 128 * QBK, *sB = tile(0, 1) + (size_t)wn * 128 * QBK;
#pragma unroll
        for (int j = 0; j < 8; j++) b0[j] = frag(sB, j * 16, 0);
#pragma unroll
        for (int i = 0; i < 8; i++) a0[i] = frag(sA, i * 16, 0);
    }
    for (int kt = 0; kt < nk; kt++) {
        const int st = kt & 1;
        const _Float16 *sA = tile(st, 0) + (size_t)wm * 128 * QBK, *sB = tile(st, 1) + (size_t)wn * 128 * QBK;
#pragma unroll
        for (int j = 0; j < 8; j++) b1[j] = frag(sB, j * 16, 1);
#pragma unroll
        for (int i = 0; i < 8; i++) a1[i] = frag(sA, i * 16, 1);
        // tiles are computed TRANSPOSED (W fragment as the MFMA's A operand): a lane ends up with 4 consecutive output columns
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0[j], a0[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[j], a1[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
        __builtin_amdgcn_sched_barrier(0);           // nothing moves across: the barrier stays three quarters into the K-step
        __syncthreads();
        if (kt + 2 < nk) issue(kt + 2, st);
        {
            const int sn = kt + 1 < nk ? st ^ 1 : st;   // (last K-step: a harmless re-read of the current stage)
            const _Float16 *nA = tile(sn, 0) + (size_t)wm * 128 * QBK, *nB = tile(sn, 1) + (size_t)wn * 128 * QBK;
#pragma unroll
            for (int j = 0; j < 8; j++) b0[j] = frag(nB, j * 16, 0);
#pragma unroll
            for (int i = 0; i < 8; i++) a0[i] = frag(nA, i * 16, 0);
        }
#pragma unroll
        for (int i = 4; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[j], a1[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 16, 1);      // the 16 global_load_lds first (VMEM)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 1);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = m0 + wm * 128 + i * 16 + (lane & 15);
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int col = n0 + wn * 128 + j * 16 + (lane >> 4) * 4;
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

template <int EPI>
void launch4w(const void *A, const void *W, void *out, int M, int N, int K, hipStream_t s) {
    constexpr int LDS = 2 * 2 * QBM * QBK * 2;
    static bool once[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !once[dev]) {
        (void)hipFuncSetAttribute((const void *)t5_gemm4w_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        once[dev] = true;
    }
    const int nn = (N + QBN - 1) / QBN, nm = (M + QBM - 1) / QBM, per_xcd = ((nm + 7) / 8 + QXM - 1) / QXM * QXM;
    hipLaunchKernelGGL(t5_gemm4w_kernel<EPI>, dim3((unsigned)(8 * per_xcd * nn)), dim3(256), LDS, s, (const _Float16 *)A, (const _Float16 *)W, out, M, N, K);
}
}  // namespace

void t5_gemm4w(int epi, const void *A, const void *W, void *out, int M, int N, int K, hipStream_t s) {
    if (epi == 0) launch4w<0>(A, W, out, M, N, K, s);
    else if (epi == 1) launch4w<1>(A, W, out, M, N, K, s);
    else launch4w<2>(A, W, out, M, N, K, s);
}

}  // namespace uc
