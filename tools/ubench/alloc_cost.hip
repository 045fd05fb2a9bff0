// tools/ubench/alloc_cost.hip — what hipMalloc / hipFree cost on this box by size, first use and re-use after a free, and the same through a
// stream-ordered pool (build: hipcc --offload-arch=gfx950 -O2 -o /tmp/alloc_cost tools/ubench/alloc_cost.hip; run: /tmp/alloc_cost)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    double t0 = now();
    (void)hipFree(nullptr);
    printf("context            %8.1f ms\n", (now() - t0) * 1e3);
    const size_t MiB = 1ull << 20;
    const size_t sizes[] = {16 * MiB, 64 * MiB, 256 * MiB, 1024 * MiB, 4096 * MiB, 16384 * MiB, 65536 * MiB};
    for (int pass = 0; pass < 2; pass++)
        for (size_t sz : sizes) {
            const int n = (int)std::max<size_t>(1, std::min<size_t>(32, (64ull << 30) / sz));   // <= 64 GiB in flight
            std::vector<void *> p(n, nullptr);
            t0 = now();
            for (int i = 0; i < n; i++)
                if (hipMalloc(&p[i], sz) != hipSuccess) { printf("hipMalloc failed at %d x %zu MiB\n", i, sz / MiB); return 1; }
            const double ta = now() - t0;
            t0 = now();
            for (int i = 0; i < n; i++) (void)hipMemsetAsync(p[i], 1, sz, 0);
            (void)hipDeviceSynchronize();
            const double ts = now() - t0;
            t0 = now();
            for (int i = 0; i < n; i++) (void)hipFree(p[i]);
            const double tf = now() - t0;
            printf("pass %d: %3d x %6zu MiB  hipMalloc %9.2f ms each (%7.1f ms per GiB)  memset %8.1f ms total  hipFree %7.2f ms each\n", pass, n, sz / MiB, ta * 1e3 / n,
                   ta * 1e3 / n / ((double)sz / (1ull << 30)), ts * 1e3, tf * 1e3 / n);
        }
    // stream-ordered pool that keeps what it is given back
    hipMemPool_t pool;
    (void)hipDeviceGetDefaultMemPool(&pool, 0);
    unsigned long long thr = ~0ull;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    for (int pass = 0; pass < 2; pass++)
        for (size_t sz : {64 * MiB, 1024 * MiB, 4096 * MiB}) {
            const int n = 16;
            std::vector<void *> p(n, nullptr);
            t0 = now();
            for (int i = 0; i < n; i++) (void)hipMallocAsync(&p[i], sz, 0);
            (void)hipStreamSynchronize(0);
            const double ta = now() - t0;
            t0 = now();
            for (int i = 0; i < n; i++) (void)hipFreeAsync(p[i], 0);
            (void)hipStreamSynchronize(0);
            printf("pool pass %d: %3d x %6zu MiB  hipMallocAsync %9.2f ms each  hipFreeAsync %7.2f ms each\n", pass, n, sz / MiB, ta * 1e3 / n, (now() - t0) * 1e3 / n);
        }
    // one slab, then a second one of the same size after freeing it
    for (int round = 0; round < 4; round++) {
        void *p = nullptr;
        t0 = now();
        if (hipMalloc(&p, 200ull << 30) != hipSuccess) { printf("200 GiB failed\n"); break; }
        const double ta = now() - t0;
        t0 = now();
        (void)hipMemset(p, 1, 200ull << 30);
        (void)hipDeviceSynchronize();
        const double ts = now() - t0;
        t0 = now();
        (void)hipFree(p);
        printf("slab round %d: 200 GiB hipMalloc %8.1f ms, memset %7.1f ms, hipFree %6.1f ms\n", round, ta * 1e3, ts * 1e3, (now() - t0) * 1e3);
    }
    return 0;
}
