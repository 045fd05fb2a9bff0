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

#include <cstdint>

#include "uc_t5.h"

namespace uc {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------- GEMM
// out[M, N] (+)= A[M, K] . W[N, K]^T      A, W f16 row-major (K contiguous: the layout of a torch Linear weight)
// EPI 0: f16 store, 1: ReLU + f16 store, 2: fp32 accumulate into out (the residual stream)
constexpr int GBM = 128, GBN = 128, GBK = 64, GLD = 72;   // LDS row stride in halves: 144 B keeps the 16-byte fragment reads of 16 rows on distinct banks

template <int EPI>
__global__ void __launch_bounds__(256) t5_gemm_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, void *__restrict__ out,
                                                      int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) _Float16 sA[GBM * GLD];
    __shared__ __attribute__((aligned(16))) _Float16 sB[GBN * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 ra[4], rb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
            ra[i] = (m0 + row < M) ? *(const uint4 *)(A + (size_t)(m0 + row) * K + (size_t)kt * GBK + kc) : uint4{0, 0, 0, 0};
            rb[i] = (n0 + row < N) ? *(const uint4 *)(W + (size_t)(n0 + row) * K + (size_t)kt * GBK + kc) : uint4{0, 0, 0, 0};
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
            *(uint4 *)(sA + row * GLD + kc) = ra[i];
            *(uint4 *)(sB + row * GLD + kc) = rb[i];
        }
    };
    const int nk = K / GBK;
    gload(0);
    sstore();
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) gload(kt + 1);          // the next tile travels while this one is multiplied
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            half8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; i++) af[i] = *(const half8 *)(sA + (wm * 64 + i * 16 + (lane & 15)) * GLD + ks * 32 + (lane >> 4) * 8);
#pragma unroll
            for (int j = 0; j < 4; j++) bf[j] = *(const half8 *)(sB + (wn * 64 + j * 16 + (lane & 15)) * GLD + ks * 32 + (lane >> 4) * 8);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) { sstore(); __syncthreads(); }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int col = n0 + wn * 64 + j * 16 + (lane & 15);
                if (col >= N) continue;
                float v = acc[i][j][r];
                if (EPI == 2) ((float *)out)[(size_t)row * N + col] += v;
                else {
                    if (EPI == 1) v = v > 0.f ? v : 0.f;
                    ((_Float16 *)out)[(size_t)row * N + col] = (_Float16)v;
                }
            }
        }
}

void t5_gemm(int epi, const void *A, const void *W, void *out, int M, int N, int K, hipStream_t s) {
    if (M <= 0) return;
    const dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM);
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
// One workgroup = one (query tile of 64 rows, head) of one sequence; wave w owns query rows 16w .. 16w+15.  d_kv = 128.
// qkv: [T, 3 * H * 128] f16 (q | k | v), out: [T, H * 128] f16.  bias: [H][2 * bias_span - 1] fp32, entry (key - query) +
// bias_span - 1.  T5 applies NO 1/sqrt(d) scaling.
constexpr int ADK = 128, AKT = 32;                      // head dim, keys per tile
constexpr int SK_LD = ADK + 8, SV_LD = AKT + 8, SP_LD = AKT + 8;

__global__ void __launch_bounds__(256) t5_attention_kernel(const _Float16 *__restrict__ qkv, const T5AttnTile *__restrict__ tiles, const float *__restrict__ bias,
                                                           int bias_span, int H, _Float16 *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) _Float16 sK[AKT * SK_LD];
    __shared__ __attribute__((aligned(16))) _Float16 sVt[ADK * SV_LD];
    __shared__ __attribute__((aligned(16))) _Float16 sP[4 * 16 * SP_LD];
    const T5AttnTile tl = tiles[blockIdx.x];
    const int h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int L = tl.len, q0 = tl.q0;
    const size_t ld = (size_t)3 * H * ADK;
    const _Float16 *qb = qkv + (size_t)tl.tok0 * ld + (size_t)h * ADK;
    const _Float16 *kb = qb + (size_t)H * ADK, *vb = qb + (size_t)2 * H * ADK;
    const float *bh = bias + (size_t)h * (2 * bias_span - 1) + (bias_span - 1);
    // this lane's A fragments of Q: row 16w + (lane & 15), dims kk * 32 + (lane >> 4) * 8 ..
    half8 qf[4];
    {
        const int qi = q0 + 16 * w + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            if (qi < L) qf[kk] = *(const half8 *)(qb + (size_t)qi * ld + kk * 32 + (lane >> 4) * 8);
            else qf[kk] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    f32x4 o[8];
#pragma unroll
    for (int d = 0; d < 8; d++) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrow[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, lrow[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += AKT) {
        // K tile row-major, V tile transposed (dims x keys): both become 16-byte B fragments
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int c = tid + 256 * i, key = c >> 4, dc = (c & 15) * 8;
            uint4 kv = {0, 0, 0, 0};
            half8 vv = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (k0 + key < L) {
                kv = *(const uint4 *)(kb + (size_t)(k0 + key) * ld + dc);
                vv = *(const half8 *)(vb + (size_t)(k0 + key) * ld + dc);
            }
            *(uint4 *)(sK + key * SK_LD + dc) = kv;
#pragma unroll
            for (int e = 0; e < 8; e++) sVt[(dc + e) * SV_LD + key] = vv[e];
        }
        __syncthreads();
        f32x4 sacc[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            sacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const half8 kf = *(const half8 *)(sK + (nt * 16 + (lane & 15)) * SK_LD + kk * 32 + (lane >> 4) * 8);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[kk], kf, sacc[nt], 0, 0, 0);
            }
        }
        // bias, mask, online softmax: this lane holds rows (lane >> 4) * 4 + r, columns nt * 16 + (lane & 15)
        float p[2][4], alpha[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int qi = q0 + 16 * w + (lane >> 4) * 4 + r;
            float mx = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const int kj = k0 + nt * 16 + (lane & 15);
                float sv = -INFINITY;
                if (kj < L) sv = sacc[nt][r] + bh[kj - min(qi, L - 1)];
                p[nt][r] = sv;
                mx = fmaxf(mx, sv);
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
            const float mnew = fmaxf(mrow[r], mx);
            alpha[r] = __expf(mrow[r] - mnew);            // first tile: exp(-inf) = 0
            float sum = 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; nt++) { p[nt][r] = __expf(p[nt][r] - mnew); sum += p[nt][r]; }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) sum += __shfl_xor(sum, m, 64);
            lrow[r] = lrow[r] * alpha[r] + sum;
            mrow[r] = mnew;
        }
        // P: C layout -> A layout through this wave's LDS slice
        _Float16 *pw = sP + w * 16 * SP_LD;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) pw[((lane >> 4) * 4 + r) * SP_LD + nt * 16 + (lane & 15)] = (_Float16)p[nt][r];
        __syncthreads();
        const half8 pf = *(const half8 *)(pw + (lane & 15) * SP_LD + (lane >> 4) * 8);
#pragma unroll
        for (int d = 0; d < 8; d++) {
#pragma unroll
            for (int r = 0; r < 4; r++) o[d][r] *= alpha[r];
            const half8 vf = *(const half8 *)(sVt + (d * 16 + (lane & 15)) * SV_LD + (lane >> 4) * 8);
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vf, o[d], 0, 0, 0);
        }
        __syncthreads();
    }
    _Float16 *ob = out + (size_t)tl.tok0 * ((size_t)H * ADK) + (size_t)h * ADK;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int qi = q0 + 16 * w + (lane >> 4) * 4 + r;
        if (qi >= L) continue;
        const float inv = 1.f / lrow[r];
#pragma unroll
        for (int d = 0; d < 8; d++) ob[(size_t)qi * ((size_t)H * ADK) + d * 16 + (lane & 15)] = (_Float16)(o[d][r] * inv);
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
                                                         const float *__restrict__ b1, float *__restrict__ h1, int T, int C1, int KW) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)T * C1; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i / C1), c = (int)(i % C1), s = seq_of[t], lo = seq_off[s] + 1, hi = seq_off[s + 1];
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
                 uint8_t *codes, float *logits, int T, int C1, int KW, int NO, hipStream_t s) {
    if (T <= 0) return;
    hipLaunchKernelGGL(t5_conv_h1_kernel, dim3((unsigned)std::min<size_t>(((size_t)T * C1 + 255) / 256, 65535)), dim3(256), 0, s, (const _Float16 *)y, ldy, seq_of, seq_off, b1, h1, T, C1, KW);
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
