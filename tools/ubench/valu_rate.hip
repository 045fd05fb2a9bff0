// valu_rate.hip — issue-rate microbenchmark for the integer VALU ops the gapped DP kernels are made of
// (gfx950).  Each wave runs ITER x 32 independent ops of one kind on 8 accumulators; prints cycles per
// wave-instruction per SIMD at full occupancy.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmax(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ uint32_t pksub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ uint32_t pkadd(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }

template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 77;
    uint32_t b = seed | 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) a[i] = (uint32_t)max((int)a[i], (int)(b + i + u * 8 + it)) ^ 1u; // v_max_i32 + v_xor
                if (KIND == 1) a[i] = pkmax(a[i], b + i + u * 8 + it) ^ 1u;                // v_pk_max_u16 + v_xor
                if (KIND == 2) a[i] = pksub(a[i], b);                                     // v_pk_sub_u16 clamp
                if (KIND == 3) a[i] = pkadd(a[i], b);                                     // v_pk_add_u16 clamp
                if (KIND == 4) a[i] = __builtin_amdgcn_perm(a[i], b, 0x0c010c00u + i);    // v_perm_b32
                if (KIND == 5) a[i] = (uint32_t)__builtin_amdgcn_sdot4((int)a[i], 0x100, (int)b, false);  // v_dot4
                if (KIND == 6) a[i] = __builtin_elementwise_sub_sat(a[i], b);             // v_sub_u32 clamp
                if (KIND == 7) a[i] = (uint32_t)max(max((int)a[i], (int)(b + it)), (int)(b + i + u)) ^ 1u; // v_max3_i32 + v_xor
                if (KIND == 9) a[i] = (a[i] + b) ^ 1u;                                   // v_add + v_xor (baseline pair)
                if (KIND == 8) a[i] = (a[i] << 5) | (b & 31);                             // v_lshl_or_b32
            }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, uint32_t *d) {
    const int iters = 20000, blocks = 256 * 8;   // 8 blocks/CU = 32 waves/CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 32;        // wave-instructions
    const double per_simd = insts / 1024.0;
    printf("%-22s %8.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<9>("v_add+v_xor", d); run<0>("v_max_i32+v_xor", d); run<7>("v_max3_i32+v_xor", d); run<6>("v_sub_u32 clamp", d); run<8>("v_lshl_or_b32", d); run<5>("v_dot4_i32_i8", d);
    run<1>("v_pk_max_u16+v_xor", d); run<2>("v_pk_sub_u16 clamp", d); run<3>("v_pk_add_u16 clamp", d); run<4>("v_perm_b32", d);
    return 0;
}
