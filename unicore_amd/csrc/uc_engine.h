// uc_engine.h — the per-GPU engine: sequence DB resident in HBM + the staged pipeline E1..E6.
// One engine = one HIP device = one process rank in the multi-GPU layout.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "uc_common.h"
#include "uc_db.h"
#include "uc_device.h"
#include "uc_options.h"
#include "unicore_cluster.h"

namespace uc {

// Memory pressure.  Both stages keep their work buffers between calls (re-allocating tens of GB per step costs more than the step at small
// sizes), and each sizes its big batches by what is free — but the fixed allocations of one stage can still meet the other stage's leftovers
// (the default workflow at 1000+ proteomes: 100+ GB of traceback matrices from the pre-step's gapped stage, then the next round's prefilter).
// A hipMalloc that fails with out-of-memory therefore asks the engine at work on this thread to give back what it is NOT using right now
// (the other stage's scratch, buffers a destroyed engine parked on the device) and is tried once more.  Engine::PressureScope registers the
// handler for the duration of a stage.
struct OomRelief { bool (*fn)(void *) = nullptr; void *ctx = nullptr; };
inline OomRelief &oom_relief_slot() { static thread_local OomRelief r; return r; }
// test hook (tests/test_gpu_parity.py::test_out_of_memory_relief): UC_TEST_OOM_AT=k makes the k-th allocation made under a registered handler
// report out-of-memory once, so the relief-and-retry path runs without 288 GB having to be filled first
inline bool oom_test_fires() {
    static std::atomic<int> left([] {
        const char *s = getenv("UC_TEST_OOM_AT");
        const int k = s ? atoi(s) : 0;
        if (k > 0) fprintf(stderr, "unicore-cluster: UC_TEST_OOM_AT=%d is set: allocation #%d made under an out-of-memory handler will be FAILED on purpose (test fault injection; "
                                   "unset it for production runs)\n", k, k);      // said once, loudly: a stray variable must not go unnoticed
        return k;
    }());
    return left.load(std::memory_order_relaxed) > 0 && left.fetch_sub(1) == 1;
}
// UC_ALLOC_LOG=1 (profiling): every device allocation of 64 MiB and more with its host time, on stderr (tools/alloc_log.py sums them)
inline bool alloc_log_on() { static const bool on = getenv("UC_ALLOC_LOG") != nullptr; return on; }
inline hipError_t malloc_with_relief(void **p, size_t bytes) {
    const bool fake = oom_relief_slot().fn && oom_test_fires();
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = fake ? hipErrorOutOfMemory : hipMalloc(p, bytes);
    if (alloc_log_on() && bytes >= ((size_t)64 << 20))
        fprintf(stderr, "unicore-cluster[alloc]: %.3f GiB in %.2f ms%s\n", bytes / 1073741824.0,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), e == hipSuccess ? "" : " FAILED");
    if (e == hipErrorOutOfMemory) {
        const OomRelief r = oom_relief_slot();
        (void)hipGetLastError();
        if (r.fn && (r.fn(r.ctx) || fake)) e = hipMalloc(p, bytes);
    }
    return e;
}

// (Measured and not kept, r04: the same buffers on HIP's virtual-memory API — address space reserved once, 1 GiB physical chunks mapped behind what
// is there as a buffer grows.  In isolation 64 x hipMemCreate(1 GiB) + hipMemMap took 1.2 ms (tools/ubench/alloc_vmm.hip), but inside the workflow the
// first 126 GiB took 3.6 s — the same ~29 ms per GiB as hipMalloc: what costs is acquiring physical memory that was used before, whatever the API —
// and address reservations of this size ran into hipErrorInvalidValue.  profiles/r04/alloc_*.log.)
template <typename T>
struct DevBuf {   // plain owning device allocation (hipMalloc), grows geometrically
    T *p = nullptr;
    size_t cap = 0;
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void release() {
        if (p) {
            const auto t0 = std::chrono::steady_clock::now();
            (void)hipFree(p);
            if (alloc_log_on() && cap * sizeof(T) >= ((size_t)64 << 20))
                fprintf(stderr, "unicore-cluster[alloc]: free %.3f GiB in %.2f ms\n", cap * sizeof(T) / 1073741824.0,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        p = nullptr; cap = 0;
    }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(cap, o.cap); }
    void reserve(size_t n) {   // contents are NOT preserved
        if (n <= cap) return;
        release();
        // (r06, measured and not kept: half again as much for buffers below 4 GiB / below 1 GiB instead of an eighth - a third of the re-allocations of the 454 small
        // buffers a nominal configs[3] call re-sizes, but 7-20 GiB more resident at configs[2], which is what tipped the gapped stage over the device's edge there:
        // profiles/r06/c3_memory_edge.txt)
        size_t want = n + n / 8 + 64;
        UC_HIP(malloc_with_relief((void **)&p, want * sizeof(T)));
        cap = want;
    }
    void reserve_exact(size_t n) {   // no growth margin (the traceback-byte buffer is sized against what is free)
        if (n <= cap) return;
        release();
        UC_HIP(malloc_with_relief((void **)&p, n * sizeof(T)));
        cap = n;
    }
    // keeps the first `used` elements.  The copy runs ON `s`, behind whatever that stream still has in flight for the old
    // buffer (a null-stream hipMemcpy does not order itself against a hipStreamNonBlocking stream), and is complete on return.
    void grow_preserve(size_t n, size_t used, hipStream_t s) {
        if (n <= cap) return;
        size_t want = n + n / 2 + 64;
        T *np = nullptr;
        UC_HIP(malloc_with_relief((void **)&np, want * sizeof(T)));
        if (used && p) UC_HIP(hipMemcpyAsync(np, p, used * sizeof(T), hipMemcpyDeviceToDevice, s));
        UC_HIP(hipStreamSynchronize(s));
        if (p) (void)hipFree(p);
        p = np;
        cap = want;
    }
};

template <typename T>
struct PinnedBuf {   // page-locked host staging buffer (hipHostMalloc): D2H copies run at link speed and skip the page faults of a fresh vector
    T *p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    void reserve(size_t n) {   // contents are NOT preserved
        if (n <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        const size_t want = n + n / 8 + 64;
        UC_HIP(hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
    }
};

struct PairIn { uint32_t q, t; int32_t qe, te; };
struct PrefilterScratch;                                  // uc_prefilter.hip
void free_prefilter_scratch(PrefilterScratch *p);
void park_prefilter_scratch(PrefilterScratch *p, int device);   // keeps one set per device for the next engine of the process
PrefilterScratch *take_prefilter_scratch(int device);
PrefilterScratch *take_parked_prefilter_scratch(int device);
struct AlignScratch;                                      // uc_align.hip
void free_align_scratch(AlignScratch *p);
bool release_tb_matrices(AlignScratch *p);   // the traceback-byte buffer, unless a MODE 7 batch loop is using it right now (uc_align.hip)
void park_align_scratch(AlignScratch *p, int device);
AlignScratch *take_align_scratch(int device);
AlignScratch *take_parked_align_scratch(int device);

struct Engine {
    Params p;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the length classes of one SW pass are independent kernels: they are spread over these streams (forked from /
    // joined to `stream`) so that the tail of one class overlaps the start of the next
    static constexpr int N_AUX = 7;
    hipStream_t aux[N_AUX] = {};
    hipEvent_t ev_fork = nullptr, ev_join[N_AUX] = {};
    int n_streams = N_AUX + 1;   // UC_STREAMS=1 serializes the class kernels on the engine stream (profiling: per-kernel durations then add up to the event time)

    // host view of the DB
    HostDb hdb;
    bool have_db = false;
    std::vector<uint32_t> h_poff;   // padded device offsets (n+1)
    std::vector<uint32_t> h_len;

    // device DB
    DevBuf<uint8_t> d_s3, d_sa;
    DevBuf<uint16_t> d_lt;
    DevBuf<uint8_t> d_raw3, d_rawa;    // unpadded tracks of the database uploaded with keep_raw (sub-databases are gathered from it)
    DevBuf<uint64_t> d_rawoff;
    uint32_t raw_n = 0;
    bool raw_resident = false;   // an EMPTY database can be resident too (raw_n == 0): the cascade of a bare "-c 0.8" on it must not fail
    DevBuf<uint32_t> d_off, d_len;
    DevBuf<int8_t> d_S3, d_SA;
    DevBuf<int8_t> d_bias;             // rule UC-1/B only
    DeviceDb ddb;

    uint32_t max_len = 1;
    uint64_t evalue_residues = 0;   // search path: residue count of the target DB (0 = the whole loaded DB)

    // stage outputs.  Hit lists (E4) and alignment records (E5/E6) are device-resident, grouped by query
    // in query order; the host keeps only the per-query counts/offsets and the accepted edges.
    std::vector<uint32_t> hit_cnt;     // per query
    std::vector<uint64_t> hit_off;     // n+1
    uint64_t n_hits = 0;
    DevBuf<uint32_t> d_hq, d_ht;       // query / target per hit
    DevBuf<int32_t> d_hs, d_hd;        // ungapped score / diagonal per hit
    DevBuf<uc_aln> d_alns;             // parallel to the hit arrays
    bool alns_valid = false;
    // accepted edges (E6): appended on the DEVICE by align(); the host copy is made only when somebody asks for it (getters,
    // multi-rank gather) - a single-rank step builds the set-cover graph straight from the device list
    DevBuf<uint32_t> d_edges;          // 2 x n_edges_dev
    uint64_t n_edges_dev = 0;
    std::vector<uint32_t> edges;       // host copy, valid iff edges_on_host
    bool edges_on_host = true;
    void clear_edges() { edges.clear(); n_edges_dev = 0; edges_on_host = true; }
    const std::vector<uint32_t> &host_edges();   // downloads if needed

    uc_stats stats{};

    explicit Engine(const Params &pp, int dev);
    ~Engine();
    void upload_db(bool keep_raw = false);
    void upload_sub_db(const std::vector<uint32_t> &cur, const std::vector<uint64_t> &full_off);   // after upload_db(true)
    uint64_t plan_db_layout();
    void finish_db_install(struct Timer &tm);
    // E1-E4: index targets [tbegin,tend), match queries [qbegin,qend) (default: all) against it       (uc_prefilter.hip)
    void prefilter(uint32_t tbegin, uint32_t tend, uint32_t qbegin = 0, uint32_t qend = UINT32_MAX);
    // the same for a rank's share of a symmetric multi-rank pass (uc_multi.cpp:prefilter_cell): target shard [tbegin,tend) against its own sequences
    // and against the queries of `others` (disjoint from the shard), every pair of the latter also mirrored (the candidate of the pair the other
    // way round, diag_select_kernel); installs the merged lists of all queries touched
    void prefilter_cells(uint32_t tbegin, uint32_t tend, const std::vector<std::pair<uint32_t, uint32_t>> &others);
    void prefilter_impl(uint32_t tbegin, uint32_t tend, uint32_t qbegin, uint32_t qend, bool mirror_all);
    bool prefilter_one(uint32_t tbegin, uint32_t tend, uint32_t qbegin, uint32_t qend, bool count_sims, double density_limit = 0.0,
                       double *density_out = nullptr, uint32_t mirror_q0 = UINT32_MAX);   // one target chunk (mirror_q0: symmetric pass, uc_prefilter.hip)
    PrefilterScratch *pre = nullptr;                       // work buffers kept between prefilter calls
    AlignScratch *aln = nullptr;                           // ... and between align calls
    uint64_t last_align_hits = 0;                          // listed pairs of the last align() whose buffers `aln` still holds (0 after the set was given back)
    void drop_scratch();                                   // parks both (results stay): virtual-rank emulation
    // out-of-memory handler of a stage (see OomRelief): stage 0 = the prefilter is at work (gives back the gapped stage's scratch), 1 = the gapped
    // stage / set cover is at work (gives back the prefilter's), 2 = neither (database upload: both); parked sets of the device go in every case
    bool relieve_pressure(int stage);
    int stage_frames[3] = {0, 0, 0};                       // open PressureScopes per stage: a stage's scratch is pinned while one of its frames is on the stack (nested scopes)
    struct PressureScope {
        OomRelief saved;
        struct Ctx { Engine *e; int stage; } ctx;
        PressureScope(Engine &E, int stage) : saved(oom_relief_slot()), ctx{&E, stage} {
            E.stage_frames[stage]++;
            oom_relief_slot() = OomRelief{[](void *c) { return ((Ctx *)c)->e->relieve_pressure(((Ctx *)c)->stage); }, &ctx};
        }
        ~PressureScope() { oom_relief_slot() = saved; ctx.e->stage_frames[ctx.stage]--; }
        PressureScope(const PressureScope &) = delete;
        PressureScope &operator=(const PressureScope &) = delete;
    };
    uint64_t prefilter_chunk_residues = 96ull << 20;   // target residues per index chunk (keeps hits/query inside the LDS filter)
    void set_hits(const uint32_t *counts, const uc_hit *h, bool check_max_seqs = true);
    void get_hits(uc_hit *out) const;                        // D2H of the device hit arrays
    void get_hits_range(uint64_t begin, uint64_t k, uc_hit *out) const;
    void export_hits_dev(uint32_t *dq, uint32_t *dt, int32_t *ds, int32_t *dd) const;
    uint64_t import_hits_dev(uint64_t n, const uint32_t *dq, const uint32_t *dt, const int32_t *ds, const int32_t *dd,
                             uint32_t rank, uint32_t world);
    // list 1 (already in key order: query, score desc, target asc) merged with list 2 (sorted here unless `sorted2`), truncated to max_seqs per query, installed
    uint64_t merge_hits_dev(uint64_t n1, const uint32_t *q1, const uint32_t *t1, const int32_t *s1, const int32_t *d1,
                            uint64_t n2, const uint32_t *q2, const uint32_t *t2, const int32_t *s2, const int32_t *d2, bool sorted2, uint32_t rank, uint32_t world);
    // the installed lists regrouped by owner rank of each pair (stable; counts[world] on the host) into caller-owned device arrays
    void partition_hits_by_owner(uint32_t world, uint32_t *dq, uint32_t *dt, int32_t *ds, int32_t *dd, uint64_t *counts);
    void get_alns(uint64_t begin, uint64_t n, uc_aln *out) const;
    void finish_hit_lists();                                  // counts/offsets from the device arrays
    void align(uint32_t qbegin, uint32_t qend);
    // E7: adjacency (sort + unique of both edge directions) on the device, greedy cover on the host
    void set_cover_device(uint32_t n, const uint32_t *h_edges, uint64_t n_edges, uint32_t *assign);
    void set_cover_own_edges(uint32_t n, uint32_t *assign);   // the engine's own (device-resident) edge list
    void set_cover_graph(uint32_t n, const uint32_t *h_edges, const uint32_t *dev_edges, uint64_t n_edges, uint32_t *assign);
    // E8a: (centre, member) candidate pairs of the linear-time pre-step for the resident DB, sorted by (centre, member), unique (uc_linclust.hip)
    std::vector<uint32_t> linclust_pairs();
    uint64_t linclust_hits();                                        // ... installed as the hit lists (query = centre); returns the pair count
    std::vector<uint32_t> linclust_pairs_impl(uint64_t *install);
    // kernel-level
    void ungapped_batch(uint64_t n, const uint32_t *q, const uint32_t *t, const int32_t *diag, int32_t *out);
    void sw_batch(int mode, const std::vector<PairIn> &pairs, int32_t *score, int32_t *qe, int32_t *te);

    double timed_ms_begin();   // records ev0 on the stream
    double timed_ms_end();     // records ev1, syncs, returns elapsed ms
};

void set_cover(uint32_t n, const uint32_t *edges, uint64_t n_edges, uint32_t *assign);
void merge_hits(uint32_t n, int max_seqs, int n_parts, const uint32_t *const *counts, const uc_hit *const *hits,
                std::vector<uint32_t> &out_cnt, std::vector<uc_hit> &out_hits);
const char *last_error_cstr();

void preload_modules(int device, bool linclust);      // uc_sw.hip: forces the code objects of the cluster path onto the device (cold start)
}  // namespace uc
