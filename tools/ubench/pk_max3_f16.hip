// pk_max3_f16.hip — does v_pk_maximum3_f16 on gfx950 order non-negative 16-bit integers (< 0x7C00) like an unsigned
// integer max, including the f16-denormal range (< 0x0400)?  And what does it cost?  Exhaustive check over all
// (a, b) pairs against 64 c values per pair + issue-rate measurement.
//   hipcc --offload-arch=gfx950 -O3 pk_max3_f16.hip -o pk_max3_f16
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ uint32_t pkmax3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__global__ void check(unsigned long long *bad, unsigned long long *nan_small) {
    // a = blockIdx (0..0x7FFF), b = thread-strided over 0..0x7FFF, c = 64 spread values; halves carry different data
    const uint32_t a = blockIdx.x;
    unsigned long long nb = 0, ns = 0;
    for (uint32_t b = threadIdx.x; b < 0x8000; b += blockDim.x)
        for (uint32_t k = 0; k < 64; k++) {
            const uint32_t c = (k * 0x1F3u + b * 7u) & 0x7FFFu;
            const uint32_t a2 = (a * 3u + 1u) & 0x7FFFu, b2 = (b * 5u + 2u) & 0x7FFFu, c2 = (c * 7u + 3u) & 0x7FFFu;
            const uint32_t r = pkmax3(a | (a2 << 16), b | (b2 << 16), c | (c2 << 16));
            const uint32_t lo = max(max(a, b), c), hi = max(max(a2, b2), c2);
            const uint32_t mlo = max(max(a, b), c), mhi = hi;
            // claim 1: all operands < 0x7C00  => exact integer max
            if (mlo < 0x7C00u && (r & 0xFFFFu) != lo) nb++;
            if (mhi < 0x7C00u && (r >> 16) != hi) nb++;
            // claim 2: some operand >= 0x7C00 (inf / NaN patterns) => result >= 0x7C00 (so an overflow is never lost)
            if (mlo >= 0x7C00u && (r & 0xFFFFu) < 0x7C00u) ns++;
            if (mhi >= 0x7C00u && (r >> 16) < 0x7C00u) ns++;
        }
    if (nb) atomicAdd(bad, nb);
    if (ns) atomicAdd(nan_small, ns);
}
__global__ void __launch_bounds__(256) rate(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = (seed * (threadIdx.x + 1) + i * 77) & 0x3FFF3FFFu;
    uint32_t b = seed & 0x1FFF1FFFu;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = pkmax3(a[i], b, (uint32_t)(i + u * 8));
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    unsigned long long *d, h[2] = {0, 0};
    hipMalloc(&d, 16); hipMemset(d, 0, 16);
    hipLaunchKernelGGL(check, dim3(0x8000), dim3(256), 0, 0, d, d + 1);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("mismatches below 0x7C00: %llu   lost overflows: %llu   (of %.3g packed triples)\n", h[0], h[1], 32768.0 * 32768.0 * 64);
    uint32_t *o; hipMalloc(&o, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, o, 100, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, o, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("v_pk_maximum3_f16 %8.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz\n", ms, ms * 1e-3 * 2.4e9 / ((double)blocks * 4 * iters * 32 / 1024.0));
    return 0;
}
