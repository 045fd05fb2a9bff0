// mfma_layout.hip — checks the lane -> element maps this repository's f16 MFMA kernels assume for
// __builtin_amdgcn_mfma_f32_16x16x32_f16 on gfx950 against a host matmul with asymmetric random operands:
//   A (16 x 32, M x K): lane l holds A[l & 15][(l >> 4) * 8 + e], e = 0..7
//   B (32 x 16, K x N): lane l holds B[(l >> 4) * 8 + e][l & 15]
//   C/D (16 x 16):      lane l, reg r holds C[(l >> 4) * 4 + r][l & 15]
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_layout mfma_layout.hip ; prints "layout ok" or the first mismatch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_ __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16 *A, const _Float16 *B, float *C) {
    const int l = threadIdx.x;
    half8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = A[(l & 15) * 32 + (l >> 4) * 8 + e]; b[e] = B[((l >> 4) * 8 + e) * 16 + (l & 15)]; }
    float4_ c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
int main() {
    _Float16 hA[16 * 32], hB[32 * 16];
    float hC[256], ref[256];
    srand(7);
    for (int i = 0; i < 512; i++) { hA[i] = (_Float16)((rand() % 17 - 8) / 4.0f); hB[i] = (_Float16)((rand() % 13 - 6) / 2.0f); }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float s = 0; for (int kk = 0; kk < 32; kk++) s += (float)hA[i * 32 + kk] * (float)hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    _Float16 *dA, *dB; float *dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    for (int i = 0; i < 256; i++) if (fabsf(hC[i] - ref[i]) > 1e-3f) { printf("MISMATCH at (%d,%d): %f vs %f\n", i / 16, i % 16, hC[i], ref[i]); return 1; }
    printf("layout ok\n");
    return 0;
}
