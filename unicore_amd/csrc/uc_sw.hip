// uc_sw.hip — dispatch for the gapped DP kernel classes, and stage E3
// (ungapped diagonal score).
#include "uc_sw_impl.hpp"

namespace uc {

void launch_sw_class_m0(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_class_m1(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_class_m2(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_class_m3(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);

// smallest (G,R) class whose G*R rows hold the query; false if it needs the generic kernel
bool sw_class_for(int lq, int *G, int *R) {
    static const int cls[][2] = {{16, 4}, {16, 8}, {16, 12}, {16, 16}, {16, 20}, {16, 24}, {16, 28}, {16, 32},
                                 {32, 20}, {32, 24}, {32, 28}, {32, 32}, {64, 20}, {64, 24}, {64, 28}, {64, 32}};
    for (const auto &c : cls)
        if (c[0] * c[1] >= lq) { *G = c[0]; *R = c[1]; return true; }
    return false;
}

void launch_sw_class(int G, int R, int mode, const SwArgs &a, uint32_t n_tasks, hipStream_t s) {
    if (n_tasks == 0) return;
    if (mode == 0) launch_sw_class_m0(G, R, a, n_tasks, s);
    else if (mode == 1) launch_sw_class_m1(G, R, a, n_tasks, s);
    else if (mode == 2) launch_sw_class_m2(G, R, a, n_tasks, s);
    else launch_sw_class_m3(G, R, a, n_tasks, s);
}

void launch_sw_pk_class_m0(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_pk_class_m1(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_pk_class_m2(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_pk_class_m4(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_pk_class_m6(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_pk_class_m7(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_pk_class(int G, int R, int mode, const SwArgs &a, uint32_t n_tasks, hipStream_t s) {
    if (n_tasks == 0) return;
    if (mode == 0) launch_sw_pk_class_m0(G, R, a, n_tasks, s);
    else if (mode == 1) launch_sw_pk_class_m1(G, R, a, n_tasks, s);
    else if (mode == 2) launch_sw_pk_class_m2(G, R, a, n_tasks, s);
    else if (mode == 4) launch_sw_pk_class_m4(G, R, a, n_tasks, s);
    else if (mode == 6) launch_sw_pk_class_m6(G, R, a, n_tasks, s);
    else launch_sw_pk_class_m7(G, R, a, n_tasks, s);
}

// ---- device layout of the database (Engine::upload_db) ----
__global__ void __launch_bounds__(256) db_pad_kernel(uint32_t n, const uint32_t *off, const uint32_t *len, const uint32_t *cur, const uint64_t *roff, const uint8_t *r3,
                                                     const uint8_t *ra, uint64_t total, uint8_t *s3, uint8_t *sa, uint16_t *lt) {
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < total + 16; p += (uint64_t)gridDim.x * 256) {
        if (p < 16) { lt[p] = (uint16_t)SW_PADPACK; continue; }     // the PAD pairs in front of the stream
        const uint64_t q = p - 16;
        uint8_t c3 = 20, ca = 20;
        uint16_t pair = (uint16_t)SW_PADPACK;
        if (n) {
            uint32_t lo = 0, hi = n;                                 // last sequence starting at or before q
            while (hi - lo > 1) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (off[mid] <= q) lo = mid; else hi = mid;
            }
            const uint64_t j = q - off[lo];
            if (j < len[lo]) {
                const uint64_t src = roff[cur ? cur[lo] : lo] + j;
                c3 = r3[src]; ca = ra[src];
                pair = (uint16_t)(c3 | (ca << 8));
            }
        }
        s3[q] = c3; sa[q] = ca; lt[p] = pair;
    }
}

void launch_db_pad(uint32_t n, const uint32_t *off, const uint32_t *len, const uint32_t *cur, const uint64_t *roff, const uint8_t *r3, const uint8_t *ra,
                   uint64_t total, uint8_t *s3, uint8_t *sa, uint16_t *lt, hipStream_t s) {
    const uint64_t work = total + 16;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((work + 255) / 256, 1u << 20);
    hipLaunchKernelGGL(db_pad_kernel, dim3(blocks), dim3(256), 0, s, n, off, len, cur, roff, r3, ra, total, s3, sa, lt);
}

// ---- rule UC-1/B (optional, default off): compositional bias per residue of the 3Di track: bias_i = round_half_away(scale * (rowsum(q_i) / 20 -
// sum over the +-20 window without i of S3[q_i][q_j] / window length)), uniform background, exact integer arithmetic ----
__global__ void __launch_bounds__(256) comp_bias_kernel(const DeviceDb db, int scale_milli, int8_t *out) {
    __shared__ int8_t S[21 * 21 + 3];
    __shared__ int rowsum[21];
    for (int i = threadIdx.x; i < 441; i += 256) S[i] = db.S3[i];
    __syncthreads();
    if (threadIdx.x < 21) { int r = 0; for (int a = 0; a < 20; a++) r += S[threadIdx.x * 21 + a]; rowsum[threadIdx.x] = r; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s < db.n; s += gridDim.x * 4) {      // a wave per sequence
        const uint8_t *q3 = db.s3 + db.off[s];
        const int lq = (int)db.len[s];
        for (int i = lane; i < lq; i += 64) {
            const int lo = i - 20 > 0 ? i - 20 : 0, hi = i + 20 < lq ? i + 20 : lq, wl = hi - lo;
            const int8_t *row = S + q3[i] * 21;
            int sum = 0;
            for (int j = lo; j < hi; j++) sum += row[q3[j]];
            sum -= row[q3[i]];
            const long long num = (long long)scale_milli * ((long long)rowsum[q3[i]] * wl - 20LL * sum), den = 20000LL * wl;
            const long long b = num >= 0 ? (num + den / 2) / den : -((-num + den / 2) / den);
            out[db.off[s] + i] = (int8_t)(b > 127 ? 127 : b < -128 ? -128 : b);
        }
    }
}
void launch_comp_bias(const DeviceDb &db, int scale_milli, int8_t *out, hipStream_t s) {
    if (db.n == 0) return;
    hipLaunchKernelGGL(comp_bias_kernel, dim3(std::min<uint32_t>((db.n + 3) / 4, 16384u)), dim3(256), 0, s, db, scale_milli, out);
}

// ---- stage E3: ungapped diagonal score (MMseqs2 UngappedAlignment on the 3Di track) ----
// One lane per candidate (q, t, diag): Kadane along the whole diagonal, saturating at 255.  The 21x21
// 3Di matrix sits in LDS; candidates arrive sorted by (q, t) so neighbouring lanes share the query.
__global__ void __launch_bounds__(256) ungapped_kernel(const DeviceDb db, uint64_t n, const uint32_t *q,
                                                       const uint32_t *t, const int32_t *diag, int32_t *score,
                                                       unsigned long long *overlap_sum) {
    __shared__ int8_t S[21 * 21 + 3];
    unsigned long long ovl = 0;
    for (int i = threadIdx.x; i < 441; i += 256) S[i] = db.S3[i];
    __syncthreads();
    for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < n; c += (uint64_t)gridDim.x * 256) {
        const uint32_t qq = q[c], tt = t[c];
        const int d = diag[c];
        const int lq = (int)db.len[qq], lt = (int)db.len[tt];
        const uint8_t *q3 = db.s3 + db.off[qq], *t3 = db.s3 + db.off[tt];
        const int i0 = d > 0 ? d : 0, i1 = min(lq, lt + d);
        int run = 0, best = 0;
        int i = i0;
        // four residues per pair of (possibly unaligned) dword loads; every sequence is followed by >= 16 pad bytes
        if (db.bias) {      // rule UC-1/B: the query position's compositional bias joins every score of its row (wave-uniform branch)
            const int8_t *qb = db.bias + db.off[qq];
            for (; i < i1; i++) {
                run = max(run + S[q3[i] * 21 + t3[i - d]] + qb[i], 0);
                best = max(best, run);
            }
        }
        // sixteen residues per pair of (unaligned) 16-byte loads: every lane follows its own diagonal, so a wave's load touches 64 lines
        // whatever its width — the kernel is bound by those line accesses, and a quarter as many loads cover the same residues (r04)
        for (; i + 16 <= i1; i += 16) {
            uint32_t wq[4], wt[4];
            __builtin_memcpy(wq, q3 + i, 16);
            __builtin_memcpy(wt, t3 + (i - d), 16);
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    run = max(run + S[((wq[w] >> (8 * b)) & 0xffu) * 21 + ((wt[w] >> (8 * b)) & 0xffu)], 0);
                    best = max(best, run);
                }
        }
        for (; i + 4 <= i1; i += 4) {
            uint32_t wq, wt;
            __builtin_memcpy(&wq, q3 + i, 4);
            __builtin_memcpy(&wt, t3 + (i - d), 4);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                run = max(run + S[((wq >> (8 * b)) & 0xffu) * 21 + ((wt >> (8 * b)) & 0xffu)], 0);
                best = max(best, run);
            }
        }
        for (; i < i1; i++) {
            run = max(run + S[q3[i] * 21 + t3[i - d]], 0);
            best = max(best, run);
        }
        score[c] = min(best, 255);
        ovl += (unsigned long long)max(i1 - i0, 0);
    }
    if (overlap_sum) {   // algorithmic-byte accounting of stage E3: one atomic per wave
        for (int o = 32; o > 0; o >>= 1) ovl += __shfl_down(ovl, o, 64);
        if ((threadIdx.x & 63) == 0 && ovl) atomicAdd(overlap_sum, ovl);
    }
}

void launch_ungapped(const DeviceDb &db, uint64_t n, const uint32_t *q, const uint32_t *t, const int32_t *diag,
                     int32_t *score, unsigned long long *overlap_sum, hipStream_t s) {
    if (n == 0) return;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(ungapped_kernel, dim3((uint32_t)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, db, n, q, t, diag, score, overlap_sum);
}

// Module preloading (cold start of a one-shot process, tools/cold_stamps.sh): HIP uploads a translation unit's code object when one of its kernels is first
// used - tens of milliseconds each for the SW classes, paid in the middle of the first pass.  A helper thread touches one kernel per module while the main
// thread builds the engine, uploads the database and runs the prefilter.
void preload_sw_pk_m0(); void preload_sw_pk_m1(); void preload_sw_pk_m4(); void preload_sw_pk_m6();
void preload_prefilter_module(); void preload_align_module(); void preload_linclust_module();
void preload_modules(int device, bool linclust) {
    if (hipSetDevice(device) != hipSuccess) return;
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)db_pad_kernel);
    preload_prefilter_module();
    if (linclust) preload_linclust_module();
    preload_align_module();
    preload_sw_pk_m0(); preload_sw_pk_m1(); preload_sw_pk_m6(); preload_sw_pk_m4();
}
}  // namespace uc
