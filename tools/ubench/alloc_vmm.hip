// tools/ubench/alloc_vmm.hip — fresh device memory three ways, on memory this process has dirtied before: N x 1 GiB hipMalloc, one N GiB hipMalloc,
// N x 1 GiB physical chunks (hipMemCreate) mapped into one reserved address range (build: hipcc --offload-arch=gfx950 -O2 -o /tmp/alloc_vmm tools/ubench/alloc_vmm.hip)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(unsigned long long *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i;
}
int main(int argc, char **argv) {
    const size_t GiB = 1ull << 30, N = argc > 1 ? (size_t)atol(argv[1]) : 64;
    CK(hipFree(nullptr));
    double t0;
    {   // dirty 200 GiB
        void *p = nullptr;
        t0 = now();
        CK(hipMalloc(&p, 200 * GiB));
        const double ta = now() - t0;
        t0 = now();
        touch<<<4096, 256>>>((unsigned long long *)p, 200 * GiB / 8);
        CK(hipDeviceSynchronize());
        printf("dirtying: 200 GiB hipMalloc %.1f ms, touch kernel %.1f ms\n", ta * 1e3, (now() - t0) * 1e3);
        CK(hipFree(p));
    }
    for (int pass = 0; pass < 2; pass++) {
        {
            std::vector<void *> p(N, nullptr);
            t0 = now();
            for (size_t i = 0; i < N; i++) CK(hipMalloc(&p[i], GiB));
            const double ta = now() - t0;
            t0 = now();
            for (size_t i = 0; i < N; i++) touch<<<1024, 256>>>((unsigned long long *)p[i], GiB / 8);
            CK(hipDeviceSynchronize());
            const double tt = now() - t0;
            t0 = now();
            for (size_t i = 0; i < N; i++) CK(hipFree(p[i]));
            printf("pass %d  A: %zu x 1 GiB hipMalloc %8.1f ms, touch %7.1f ms, free %6.1f ms\n", pass, N, ta * 1e3, tt * 1e3, (now() - t0) * 1e3);
        }
        {
            void *p = nullptr;
            t0 = now();
            CK(hipMalloc(&p, N * GiB));
            const double ta = now() - t0;
            t0 = now();
            touch<<<4096, 256>>>((unsigned long long *)p, N * GiB / 8);
            CK(hipDeviceSynchronize());
            const double tt = now() - t0;
            t0 = now();
            CK(hipFree(p));
            printf("pass %d  B: 1 x %zu GiB hipMalloc %8.1f ms, touch %7.1f ms, free %6.1f ms\n", pass, N, ta * 1e3, tt * 1e3, (now() - t0) * 1e3);
        }
        {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = 0;
            size_t gran = 0;
            CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            void *va = nullptr;
            t0 = now();
            CK(hipMemAddressReserve(&va, N * GiB, gran, nullptr, 0));
            const double tr = now() - t0;
            std::vector<hipMemGenericAllocationHandle_t> h(N);
            t0 = now();
            for (size_t i = 0; i < N; i++) CK(hipMemCreate(&h[i], GiB, &prop, 0));
            const double tc = now() - t0;
            t0 = now();
            for (size_t i = 0; i < N; i++) CK(hipMemMap((char *)va + i * GiB, GiB, 0, h[i], 0));
            hipMemAccessDesc d = {};
            d.location = prop.location;
            d.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, N * GiB, &d, 1));
            const double tm = now() - t0;
            t0 = now();
            touch<<<4096, 256>>>((unsigned long long *)va, N * GiB / 8);
            CK(hipDeviceSynchronize());
            const double tt = now() - t0;
            t0 = now();
            CK(hipMemUnmap(va, N * GiB));
            for (size_t i = 0; i < N; i++) CK(hipMemRelease(h[i]));
            CK(hipMemAddressFree(va, N * GiB));
            printf("pass %d  C: VMM granularity %zu KiB: reserve %.2f ms, %zu x hipMemCreate(1 GiB) %8.1f ms, map + access %7.1f ms, touch %7.1f ms, unmap + release %6.1f ms\n", pass,
                   gran >> 10, tr * 1e3, N, tc * 1e3, tm * 1e3, tt * 1e3, (now() - t0) * 1e3);
        }
    }
    return 0;
}
