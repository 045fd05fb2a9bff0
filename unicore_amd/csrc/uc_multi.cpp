// uc_multi.cpp — multi-GPU exchange of the cluster path: grid plan, RCCL all-gather of the per-shard hit
// lists, device merge, edge gather.  See uc_multi.h for the layout.  RCCL is called from here — the library itself —
// so `unicore cluster` / `foldseek cluster` use every GPU of the node with no Python in the process.
#include "uc_multi.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#define UC_NCCL(call)                                                                                          \
    do {                                                                                                       \
        ncclResult_t _r = (call);                                                                              \
        if (_r != ncclSuccess) ::uc::fail(UC_ERR_DEVICE, "RCCL error at %s:%d: %s", __FILE__, __LINE__, ncclGetErrorString(_r)); \
    } while (0)

namespace uc {

// ---------------------------------------------------------------------------------------------- in-process group
void LocalGroup::barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (failed) fail(UC_ERR_GENERIC, "another GPU rank of this run failed");
    const uint64_t gen = generation;
    if (++waiting == world) {
        waiting = 0;
        generation++;
        cv.notify_all();
        return;
    }
    cv.wait(lk, [&] { return generation != gen || failed; });
    if (failed && generation == gen) fail(UC_ERR_GENERIC, "another GPU rank of this run failed");
}

void LocalGroup::fail_all() {
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    cv.notify_all();
}

// ---------------------------------------------------------------------------------------------- communicator
struct CommScratch {
    DevBuf<uint64_t> sz_send, sz_recv;
    DevBuf<int32_t> pad, all, cat, own, part, acc;
    DevBuf<uint32_t> e_send, e_recv;
};

Comm::Comm() : scratch(new CommScratch) {}
Comm::~Comm() {
    scratch.reset();
    if (ncclComm *h = nccl.exchange(nullptr)) (void)ncclCommDestroy(h);
}

// How long an abort waits for the threads that are inside an RCCL call with the handle it is about to reclaim.  An enqueue is host-side work that
// returns in microseconds, so in the common case the wait is over at once; what can still be inside after milliseconds is a thread blocked in a
// first-use connect (ncclGroupEnd of the first exchange) on the very peer that is dying - nothing but the abort gets that thread out, so the wait must
// be bounded.  2 s: three orders of magnitude above any healthy enqueue, and short against the run it ends (a failing N-rank call reports within
// seconds).  When it expires the abort proceeds UNDER the live enqueue by design, and says so (verbosity >= 2).
constexpr int ABORT_GRACE_MS = 2000;

void Comm::mark_aborted() { aborted.store(true); }

void Comm::abort() {
    aborted.store(true);
    // whoever takes the handle out aborts it, exactly once; no lock: the thread that owns this communicator may be blocked inside
    // ncclGroupEnd / a first-use connect waiting for the very peer that is calling us
    ncclComm *h = nccl.exchange(nullptr);
    if (!h) return;
    // ncclCommAbort reclaims the communicator: a thread that read the handle before the exchange above and is still inside its RCCL call
    // (nccl_enqueue / comm_info count themselves in BEFORE they read the handle, so they are visible here) must be out first
    int ms = 0;
    for (; ms < ABORT_GRACE_MS && in_flight.load() > 0; ms++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (in_flight.load() > 0)
        logf(2, "unicore-cluster: rank %d: %d thread(s) still inside an RCCL call after the %d ms abort grace (blocked on a dying peer): aborting the communicator under them\n",
             rank, in_flight.load(), ABORT_GRACE_MS);
    (void)ncclCommAbort(h);
}

// the failure path of an N-rank run: every communicator is MARKED first (new enqueues fail at once, the ones inside start to drain everywhere at the
// same time), then the bounded waits and aborts follow - the grace periods overlap instead of adding up to 2 N seconds
void abort_all(const std::vector<Comm *> &comms) {
    for (Comm *c : comms) if (c) c->mark_aborted();
    for (Comm *c : comms) if (c) c->abort();
}

namespace {
// ncclGroupStart ... ncclGroupEnd as a scope: an exception between the two (a failed ncclSend, an injected failure) must not leave the
// calling thread's group open — the abort that follows would run INSIDE that group
struct GroupScope {
    bool open = false;
    GroupScope() { UC_NCCL(ncclGroupStart()); open = true; }
    void end() { open = false; UC_NCCL(ncclGroupEnd()); }
    ~GroupScope() { if (open) (void)ncclGroupEnd(); }
};

// test hook: UC_FAIL_RANK="<rank>:<stage>"; stage 2 = inside the grouped point-to-point exchange, between ncclGroupStart and ncclGroupEnd
bool inject_failure(int rank, int stage) {
    int fr = -1, fs = -1;
    if (const char *e = getenv("UC_FAIL_RANK")) sscanf(e, "%d:%d", &fr, &fs);
    return fr == rank && fs == stage;
}

// RCCL enqueue: the handle is read once; an abort() that lands between the read and the call makes the call fail (or return early) — both
// surface as an error of this rank, which is what the caller wants to hear
struct InFlight {      // counted in before the handle is read, out when the call (and its GroupScope) has been left, also by an exception
    std::atomic<int> &n;
    explicit InFlight(std::atomic<int> &c) : n(c) { n.fetch_add(1); }
    ~InFlight() { n.fetch_sub(1); }
};
template <class F>
void nccl_enqueue(Comm &C, F &&f) {
    InFlight guard(C.in_flight);
    ncclComm *h = C.nccl.load();
    if (C.aborted.load() || !h) fail(UC_ERR_DEVICE, "RCCL communicator of rank %d was aborted (another GPU rank of this run failed)", C.rank);
    f(h);
    if (C.aborted.load()) fail(UC_ERR_DEVICE, "RCCL communicator of rank %d was aborted (another GPU rank of this run failed)", C.rank);
}
}  // namespace

void comm_unique_id(uint8_t id[128]) {
    static_assert(NCCL_UNIQUE_ID_BYTES == 128, "uc_comm id size");
    ncclUniqueId u;
    UC_NCCL(ncclGetUniqueId(&u));
    memcpy(id, u.internal, 128);
}

void comm_init_rank(Comm &C, const uint8_t id[128], int rank, int world, int device) {
    ncclUniqueId u;
    memcpy(u.internal, id, 128);
    UC_HIP(hipSetDevice(device));
    C.rank = rank;
    C.world = world;
    ncclComm *h = nullptr;
    UC_NCCL(ncclCommInitRank(&h, world, u, rank));
    C.nccl.store(h);
    C.uses_rccl = true;
}

void comm_info(const Comm &C, int *count, int *rank, int *device) {
    InFlight guard(const_cast<Comm &>(C).in_flight);      // the info calls read the handle too: an abort must not reclaim it under them
    ncclComm *h = C.nccl.load();
    if (!h || C.aborted.load()) fail(UC_ERR_ARGS, "communicator has no RCCL handle");
    UC_NCCL(ncclCommCount(h, count));
    UC_NCCL(ncclCommUserRank(h, rank));
    UC_NCCL(ncclCommCuDevice(h, device));
}

void comm_init_all(const std::vector<Comm *> &comms, const std::vector<int> &devices) {
    const int W = (int)comms.size();
    std::vector<ncclComm_t> h((size_t)W, nullptr);
    UC_NCCL(ncclCommInitAll(h.data(), W, devices.data()));
    for (int r = 0; r < W; r++) { comms[(size_t)r]->rank = r; comms[(size_t)r]->world = W; comms[(size_t)r]->nccl.store(h[(size_t)r]); comms[(size_t)r]->uses_rccl = true; }
}

// A communicator that owns an RCCL handle always goes through RCCL, even with one rank: a 1-rank communicator on the
// single-GPU test box exercises exactly the calls an 8-rank run makes.
void Comm::barrier(Engine &E) {
    if (world == 1 && !uses_rccl) return;
    if (grp) { grp->barrier(); return; }
    std::vector<uint64_t> dummy((size_t)world);
    all_gather_u64(E, 0, dummy.data());   // cross-process: the smallest collective doubles as the barrier
}

void Comm::all_gather_u64(Engine &E, uint64_t v, uint64_t *out) {
    if (world == 1 && !uses_rccl) { out[0] = v; return; }
    if (grp) {
        grp->val[(size_t)rank] = v;
        grp->barrier();
        for (int r = 0; r < world; r++) out[r] = grp->val[(size_t)r];
        grp->barrier();   // nobody overwrites val before everybody has read it
        return;
    }
    CommScratch &S = *scratch;
    S.sz_send.reserve(1);
    S.sz_recv.reserve((size_t)world);
    UC_HIP(hipMemcpyAsync(S.sz_send.p, &v, 8, hipMemcpyHostToDevice, E.stream));
    nccl_enqueue(*this, [&](ncclComm *h) { UC_NCCL(ncclAllGather(S.sz_send.p, S.sz_recv.p, 1, ncclUint64, h, E.stream)); });
    UC_HIP(hipMemcpyAsync(out, S.sz_recv.p, (size_t)world * 8, hipMemcpyDeviceToHost, E.stream));
    UC_HIP(hipStreamSynchronize(E.stream));
}

void Comm::all_gather_dev(Engine &E, const void *send, void *recv, size_t bytes) {
    if (world == 1 && !uses_rccl) {
        if (bytes) UC_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, E.stream));
        UC_HIP(hipStreamSynchronize(E.stream));
        return;
    }
    if (uses_rccl) {
        if (grp) grp->barrier();   // a rank that failed earlier must not leave its peers inside the collective
        nccl_enqueue(*this, [&](ncclComm *h) { UC_NCCL(ncclAllGather(send, recv, bytes, ncclUint8, h, E.stream)); });
        UC_HIP(hipStreamSynchronize(E.stream));
        if (grp) grp->barrier();   // ... and one that failed DURING it (its handler aborts every communicator) is reported as such
        return;
    }
    // in-process ranks without RCCL (several engines on ONE device — RCCL refuses two ranks per GPU): every rank
    // publishes its send buffer and copies its peers' with the copy engine
    UC_HIP(hipStreamSynchronize(E.stream));   // the send buffer is complete
    grp->ptr[(size_t)rank] = send;
    grp->barrier();
    for (int r = 0; r < world && bytes; r++)
        UC_HIP(hipMemcpyAsync((char *)recv + (size_t)r * bytes, grp->ptr[(size_t)r], bytes, hipMemcpyDefault, E.stream));
    UC_HIP(hipStreamSynchronize(E.stream));
    grp->barrier();   // peers may reuse their send buffers only now
}

void Comm::broadcast_dev(Engine &E, void *buf, size_t bytes, int root) {
    if (world == 1 && !uses_rccl) return;
    if (uses_rccl) {
        if (grp) grp->barrier();
        nccl_enqueue(*this, [&](ncclComm *h) { UC_NCCL(ncclBroadcast(buf, buf, bytes, ncclUint8, root, h, E.stream)); });
        UC_HIP(hipStreamSynchronize(E.stream));
        if (grp) grp->barrier();
        return;
    }
    UC_HIP(hipStreamSynchronize(E.stream));
    if (rank == root) grp->ptr[(size_t)root] = buf;
    grp->barrier();
    if (rank != root && bytes) UC_HIP(hipMemcpyAsync(buf, grp->ptr[(size_t)root], bytes, hipMemcpyDefault, E.stream));
    UC_HIP(hipStreamSynchronize(E.stream));
    grp->barrier();
}

uint64_t Comm::gather_edges_dev(Engine &E, const uint32_t **dev_out) {
    Engine::PressureScope ps(E, 1);      // e_send / e_recv (2 x all ranks' edges on rank 0) may ask for the prefilter's work buffers and the traceback bytes back
    CommScratch &S = *scratch;
    if (dev_out) *dev_out = nullptr;
    // this rank's list as a device array (align() leaves it on the device; a list that only exists on the host is staged once)
    const uint64_t mine = E.edges_on_host ? E.edges.size() / 2 : E.n_edges_dev;
    const uint32_t *src = E.d_edges.p;
    if (E.edges_on_host && mine) {
        S.e_send.reserve(2 * mine);
        UC_HIP(hipMemcpyAsync(S.e_send.p, E.edges.data(), 2 * mine * 4, hipMemcpyHostToDevice, E.stream));
        src = S.e_send.p;
    }
    if (world == 1 && !uses_rccl) {
        UC_HIP(hipStreamSynchronize(E.stream));
        if (dev_out) *dev_out = src;
        return mine;
    }
    std::vector<uint64_t> sz((size_t)world);
    all_gather_u64(E, mine, sz.data());
    Timer tg;   // phase 6 = the gather itself: the size exchange above is also where a rank waits for the slowest rank's gapped stage (that rank's phase 5)
    uint64_t tot = 0;
    for (uint64_t c : sz) tot += c;
    if (rank == 0) S.e_recv.reserve(std::max<uint64_t>(2 * tot, 1));
    if (uses_rccl) {
        if (grp) grp->barrier();
        nccl_enqueue(*this, [&](ncclComm *h) {
            GroupScope g;
            if (rank == 0) {
                uint64_t o = sz[0];
                for (int r = 1; r < world; r++) {
                    if (sz[(size_t)r]) UC_NCCL(ncclRecv(S.e_recv.p + 2 * o, 2 * sz[(size_t)r], ncclUint32, r, h, E.stream));
                    o += sz[(size_t)r];
                }
            } else if (mine) {
                UC_NCCL(ncclSend(src, 2 * mine, ncclUint32, 0, h, E.stream));
            }
            g.end();
        });
        if (rank == 0 && mine) UC_HIP(hipMemcpyAsync(S.e_recv.p, src, 2 * mine * 4, hipMemcpyDeviceToDevice, E.stream));
        UC_HIP(hipStreamSynchronize(E.stream));
        if (grp) grp->barrier();
    } else {
        // virtual ranks (several engines on one device): publish the list, rank 0 copies
        UC_HIP(hipStreamSynchronize(E.stream));
        grp->ptr[(size_t)rank] = src;
        grp->barrier();
        if (rank == 0) {
            uint64_t o = 0;
            for (int r = 0; r < world; r++) {
                if (sz[(size_t)r]) UC_HIP(hipMemcpyAsync(S.e_recv.p + 2 * o, grp->ptr[(size_t)r], 2 * sz[(size_t)r] * 4, hipMemcpyDefault, E.stream));
                o += sz[(size_t)r];
            }
            UC_HIP(hipStreamSynchronize(E.stream));
        }
        grp->barrier();   // the peers' lists may change only now
    }
    if (rank == 0 && dev_out) *dev_out = S.e_recv.p;
    E.stats.exchange_bytes += rank == 0 ? 8 * (tot - mine) : 0;
    E.stats.phase_seconds[6] += tg.seconds();
    return tot;
}

// ---------------------------------------------------------------------------------------------- grid plan
std::vector<std::pair<uint32_t, uint32_t>> shard_ranges(const std::vector<uint32_t> &len, int parts) {
    // contiguous ranges of ~equal residue counts (range partition by key)
    const uint32_t n = (uint32_t)len.size();
    std::vector<uint64_t> cum((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) cum[i + 1] = cum[i] + len[i];
    const uint64_t total = cum[n];
    std::vector<uint32_t> b((size_t)parts + 1, 0);
    for (int g = 1; g < parts; g++) {
        const double want = (double)total * g / parts;
        // first index whose prefix sum is >= want (== numpy.searchsorted(cum, want, side="left"))
        b[(size_t)g] = (uint32_t)(std::lower_bound(cum.begin(), cum.end(), want, [](uint64_t c, double w) { return (double)c < w; }) - cum.begin());
        b[(size_t)g] = std::min(b[(size_t)g], n);
    }
    b[(size_t)parts] = n;
    for (int g = 1; g <= parts; g++) b[(size_t)g] = std::max(b[(size_t)g], b[(size_t)g - 1]);
    std::vector<std::pair<uint32_t, uint32_t>> r;
    for (int g = 0; g < parts; g++) r.emplace_back(b[(size_t)g], b[(size_t)g + 1]);
    return r;
}

void grid_shape(int world, int target_shards, int *Q, int *T) {
    int t = target_shards > 0 ? target_shards : world;
    if (t > world || world % t) fail(UC_ERR_ARGS, "--target-shards %d must divide the number of GPUs (%d)", t, world);
    *T = t;
    *Q = world / t;
}

GridCell grid_cell(const std::vector<uint32_t> &len, int world, int target_shards, int rank) {
    int Q, T;
    grid_shape(world, target_shards, &Q, &T);
    const auto tr = shard_ranges(len, T), qr = shard_ranges(len, Q);
    return {tr[(size_t)(rank % T)].first, tr[(size_t)(rank % T)].second, qr[(size_t)(rank / T)].first, qr[(size_t)(rank / T)].second};
}

void prefilter_cell(Engine &E, int world, int target_shards, int rank) {
    int Q, T;
    grid_shape(world, target_shards, &Q, &T);
    const GridCell g = grid_cell(E.h_len, world, target_shards, rank);
    const char *off = getenv("UC_PREFILTER_SYMMETRIC");
    const bool sym = T == world && world > 1 && E.p.mat_symmetric && !(off && atoi(off) == 0);
    if (!sym) { E.prefilter(g.tb, g.te, g.qb, g.qe); return; }
    const auto shards = shard_ranges(E.h_len, world);
    std::vector<std::pair<uint32_t, uint32_t>> others;
    for (int d = 1; d <= world / 2; d++) {
        if (2 * d == world && rank >= world / 2) continue;      // the block at distance N/2 is reachable from both sides: the lower rank takes it
        others.push_back(shards[(size_t)((rank + d) % world)]);
    }
    E.prefilter_cells(g.tb, g.te, others);
}

// ---------------------------------------------------------------------------------------------- the exchange
namespace {
uint64_t round_limit() {   // records a rank receives per round of exchange 1 (beyond it the home range is worked off in several rounds)
    if (const char *e = getenv("UC_EXCHANGE_LIMIT")) return std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    return 1ull << 30;
}

}  // namespace

void Comm::all_gather_u64s(Engine &E, const uint64_t *v, int k, uint64_t *out) {
    if (world == 1 && !uses_rccl) { memcpy(out, v, (size_t)k * 8); return; }
    if (grp) {
        grp->ptr[(size_t)rank] = v;
        grp->barrier();
        for (int r = 0; r < world; r++) memcpy(out + (size_t)r * k, grp->ptr[(size_t)r], (size_t)k * 8);
        grp->barrier();   // nobody lets go of its vector before everybody has read it
        return;
    }
    CommScratch &S = *scratch;
    S.sz_send.reserve((size_t)k);
    S.sz_recv.reserve((size_t)world * k);
    UC_HIP(hipMemcpyAsync(S.sz_send.p, v, (size_t)k * 8, hipMemcpyHostToDevice, E.stream));
    nccl_enqueue(*this, [&](ncclComm *h) { UC_NCCL(ncclAllGather(S.sz_send.p, S.sz_recv.p, (size_t)k, ncclUint64, h, E.stream)); });
    UC_HIP(hipMemcpyAsync(out, S.sz_recv.p, (size_t)world * k * 8, hipMemcpyDeviceToHost, E.stream));
    UC_HIP(hipStreamSynchronize(E.stream));
}

// Ragged all-to-all of `na` parallel 4-byte arrays: rank r sends elements [send_off[p], send_off[p] + send_cnt[p]) of every
// send array to rank p and receives recv_cnt[p] elements from p at recv_off[p] of every recv array.  RCCL: ONE group of
// point-to-point sends / receives (over xGMI every pair of GPUs has its own link: the transfers of a rank run side by side);
// virtual ranks: device copies from the peers' published buffers.
void Comm::all_to_all_dev(Engine &E, int na, const void *const *send, const uint64_t *send_off, const uint64_t *send_cnt,
                          void *const *recv, const uint64_t *recv_off, const uint64_t *recv_cnt, double *t_wait, double *t_move) {
    const int me = rank;
    Timer tw;
    auto waited = [&] { if (t_wait) *t_wait += tw.seconds(); tw = Timer(); };
    auto moved = [&] { if (t_move) *t_move += tw.seconds(); tw = Timer(); };
    if (send_cnt[me] != recv_cnt[me]) fail(UC_ERR_GENERIC, "all_to_all_dev: inconsistent self segment");
    if (world == 1 && !uses_rccl) {
        for (int k = 0; k < na && send_cnt[0]; k++)
            UC_HIP(hipMemcpyAsync((char *)recv[k] + 4 * recv_off[0], (const char *)send[k] + 4 * send_off[0], 4 * send_cnt[0], hipMemcpyDeviceToDevice, E.stream));
        UC_HIP(hipStreamSynchronize(E.stream));
        moved();
        return;
    }
    if (uses_rccl) {
        if (grp) grp->barrier();   // a rank that failed earlier must not leave its peers inside the exchange
        waited();
        nccl_enqueue(*this, [&](ncclComm *h) {
            GroupScope g;
            for (int p = 0; p < world; p++) {
                if (p == me) continue;
                for (int k = 0; k < na; k++) {
                    if (send_cnt[p]) UC_NCCL(ncclSend((const char *)send[k] + 4 * send_off[p], send_cnt[p], ncclUint32, p, h, E.stream));
                    if (recv_cnt[p]) UC_NCCL(ncclRecv((char *)recv[k] + 4 * recv_off[p], recv_cnt[p], ncclUint32, p, h, E.stream));
                }
                if (p >= world / 2 && inject_failure(me, 2)) fail(UC_ERR_DEVICE, "injected failure of rank %d inside the grouped exchange (UC_FAIL_RANK)", me);
            }
            if (inject_failure(me, 2)) fail(UC_ERR_DEVICE, "injected failure of rank %d inside the grouped exchange (UC_FAIL_RANK)", me);
            g.end();
        });
        for (int k = 0; k < na && send_cnt[me]; k++)
            UC_HIP(hipMemcpyAsync((char *)recv[k] + 4 * recv_off[me], (const char *)send[k] + 4 * send_off[me], 4 * send_cnt[me], hipMemcpyDeviceToDevice, E.stream));
        UC_HIP(hipStreamSynchronize(E.stream));
        moved();
        if (grp) grp->barrier();   // ... and one that failed DURING it (its handler aborts every communicator) is reported as such
        waited();
        return;
    }
    // in-process ranks without RCCL (several engines on ONE device): publish the send side, copy from the peers
    struct Pub { const void *const *send; const uint64_t *off; } pub{send, send_off};
    UC_HIP(hipStreamSynchronize(E.stream));   // the send buffers are complete
    grp->ptr[(size_t)me] = &pub;
    grp->barrier();
    waited();
    for (int p = 0; p < world; p++) {
        if (!recv_cnt[p]) continue;
        const Pub &pp = *(const Pub *)grp->ptr[(size_t)p];
        for (int k = 0; k < na; k++)
            UC_HIP(hipMemcpyAsync((char *)recv[k] + 4 * recv_off[p], (const char *)pp.send[k] + 4 * pp.off[(size_t)me], 4 * recv_cnt[p], hipMemcpyDefault, E.stream));
    }
    UC_HIP(hipStreamSynchronize(E.stream));
    moved();
    grp->barrier();   // peers may reuse their send buffers only now
    waited();
}

// The exchange of the sharded pass, two phases (r4; it replaced "all-gather the union, sort it on every rank, keep 1/N"):
//   1. every query has a HOME rank (contiguous query ranges of equal residue counts).  A rank's lists are grouped by query, so
//      the records of one home are one contiguous slice: ragged all-to-all, then the home rank merges the slices of its queries
//      under (score desc, target asc) and truncates to max_seqs — 1/N of the union per rank instead of all of it on every rank
//      (lossless: global top-M is a subset of the union of the shard top-Ms);
//   2. the surviving pairs go to the rank that OWNS them (hash of the unordered pair's representative query: mutual hits meet on
//      one rank and share their DPs there, a rank owns whole queries): stable partition by owner on the device, second ragged
//      all-to-all, install.  Every merged pair is aligned exactly once over all ranks; the result does not depend on N or the grid.
uint64_t exchange_hits(Engine &E, Comm &C) {
    Timer tm;
    UC_HIP(hipSetDevice(E.device));
    const int W = C.world, me = C.rank;
    CommScratch &S = *C.scratch;
    const auto homes = shard_ranges(E.h_len, W);
    std::vector<uint64_t> soff((size_t)W), scnt((size_t)W), mat((size_t)W * W), roff((size_t)W), rcnt((size_t)W);

    // ---- phase 1 (in rounds if a rank would receive more than round_limit() records at once)
    for (int h = 0; h < W; h++) { soff[(size_t)h] = E.hit_off[homes[(size_t)h].first]; scnt[(size_t)h] = E.hit_off[homes[(size_t)h].second] - soff[(size_t)h]; }
    C.all_gather_u64s(E, scnt.data(), W, mat.data());
    uint64_t worst = 0;
    for (int h = 0; h < W; h++) { uint64_t c = 0; for (int r = 0; r < W; r++) c += mat[(size_t)r * W + h]; worst = std::max(worst, c); }
    const uint64_t rounds = std::max<uint64_t>(1, (worst + round_limit() - 1) / round_limit());
    double t_x1 = 0, t_merge = 0;
    uint64_t nacc = 0;     // merged records of the rounds so far (rounds > 1 only), parked in S.acc as [4][cap]
    uint64_t acc_cap = 0;
    for (uint64_t rd = 0; rd < rounds; rd++) {
        Timer tx;
        if (rounds > 1) {   // sub-range rd of every home range (by query count); counts are exchanged per round
            for (int h = 0; h < W; h++) {
                const uint64_t qb = homes[(size_t)h].first, qn = homes[(size_t)h].second - qb;
                const uint32_t a = (uint32_t)(qb + qn * rd / rounds), b = (uint32_t)(qb + qn * (rd + 1) / rounds);
                soff[(size_t)h] = E.hit_off[a]; scnt[(size_t)h] = E.hit_off[b] - E.hit_off[a];
            }
            C.all_gather_u64s(E, scnt.data(), W, mat.data());
        }
        uint64_t R = 0;
        for (int r = 0; r < W; r++) { rcnt[(size_t)r] = mat[(size_t)r * W + me]; roff[(size_t)r] = R; R += rcnt[(size_t)r]; }
        S.all.reserve(4 * std::max<uint64_t>(R, 1));
        const void *snd[4] = {E.d_hq.p, E.d_ht.p, E.d_hs.p, E.d_hd.p};
        void *rcv[4] = {S.all.p, S.all.p + R, S.all.p + 2 * R, S.all.p + 3 * R};
        C.all_to_all_dev(E, 4, snd, soff.data(), scnt.data(), rcv, roff.data(), rcnt.data());
        E.stats.exchange_bytes += 16 * (R - rcnt[(size_t)me]);
        t_x1 += tx.seconds();
        if (rounds == 1) {
            Turn turn(C, &E);
            Timer tg;      // (started inside the turn: waiting for the other virtual ranks' turns is not this rank's time)
            E.import_hits_dev(R, (uint32_t *)S.all.p, (uint32_t *)S.all.p + R, S.all.p + 2 * R, S.all.p + 3 * R, 0, 1);
            t_merge += tg.seconds();
        } else {
            // the merge installs its result IN the engine, whose own lists are still the send side of the later rounds: park them, merge
            // this round's slices, append the result to the accumulator, put the own lists back
            Turn turn(C, &E);
            Timer tg;
            const uint64_t nloc = E.n_hits;
            S.own.reserve(4 * std::max<uint64_t>(nloc, 1));
            if (nloc) E.export_hits_dev((uint32_t *)S.own.p, (uint32_t *)S.own.p + nloc, S.own.p + 2 * nloc, S.own.p + 3 * nloc);
            const std::vector<uint32_t> cnt_keep = E.hit_cnt;
            const std::vector<uint64_t> off_keep = E.hit_off;
            const uint64_t k = E.import_hits_dev(R, (uint32_t *)S.all.p, (uint32_t *)S.all.p + R, S.all.p + 2 * R, S.all.p + 3 * R, 0, 1);
            if (nacc + k > acc_cap) {   // grow the [4][cap] accumulator, keeping its rows
                const uint64_t ncap = (nacc + k) * 3 / 2 + 64;
                S.cat.reserve(4 * ncap);
                for (int a = 0; a < 4 && nacc; a++)
                    UC_HIP(hipMemcpyAsync(S.cat.p + a * ncap, S.acc.p + a * acc_cap, 4 * nacc, hipMemcpyDeviceToDevice, E.stream));
                UC_HIP(hipStreamSynchronize(E.stream));
                S.cat.swap(S.acc);
                acc_cap = ncap;
            }
            if (k) E.export_hits_dev((uint32_t *)S.acc.p + nacc, (uint32_t *)S.acc.p + acc_cap + nacc, S.acc.p + 2 * acc_cap + nacc, S.acc.p + 3 * acc_cap + nacc);
            nacc += k;
            // restore the send side (order is already the list order: the install below only rebuilds counts and offsets)
            if (rd + 1 < rounds) {
                E.d_hq.reserve(nloc); E.d_ht.reserve(nloc); E.d_hs.reserve(nloc); E.d_hd.reserve(nloc);
                UC_HIP(hipMemcpyAsync(E.d_hq.p, S.own.p, 4 * nloc, hipMemcpyDeviceToDevice, E.stream));
                UC_HIP(hipMemcpyAsync(E.d_ht.p, S.own.p + nloc, 4 * nloc, hipMemcpyDeviceToDevice, E.stream));
                UC_HIP(hipMemcpyAsync(E.d_hs.p, S.own.p + 2 * nloc, 4 * nloc, hipMemcpyDeviceToDevice, E.stream));
                UC_HIP(hipMemcpyAsync(E.d_hd.p, S.own.p + 3 * nloc, 4 * nloc, hipMemcpyDeviceToDevice, E.stream));
                UC_HIP(hipStreamSynchronize(E.stream));
                E.n_hits = nloc; E.hit_cnt = cnt_keep; E.hit_off = off_keep;
            } else {
                E.import_hits_dev(nacc, (uint32_t *)S.acc.p, (uint32_t *)S.acc.p + acc_cap, S.acc.p + 2 * acc_cap, S.acc.p + 3 * acc_cap, 0, 1);
            }
            t_merge += tg.seconds();
        }
    }
    E.stats.phase_seconds[1] += t_x1;
    E.stats.phase_seconds[2] += t_merge;

    // ---- phase 2: pairs to their owners
    const uint64_t K = E.n_hits;
    S.pad.reserve(4 * std::max<uint64_t>(K, 1));
    double t_part = 0;
    {
        Turn turn(C, &E);
        Timer tp;
        E.partition_hits_by_owner((uint32_t)W, (uint32_t *)S.pad.p, (uint32_t *)S.pad.p + K, S.pad.p + 2 * K, S.pad.p + 3 * K, scnt.data());
        t_part = tp.seconds();
    }
    Timer t3;
    uint64_t o = 0;
    for (int p = 0; p < W; p++) { soff[(size_t)p] = o; o += scnt[(size_t)p]; }
    C.all_gather_u64s(E, scnt.data(), W, mat.data());
    const double t_counts = t3.seconds();
    double t_wait = 0, t_move = 0;
    uint64_t R2 = 0;
    for (int r = 0; r < W; r++) { rcnt[(size_t)r] = mat[(size_t)r * W + me]; roff[(size_t)r] = R2; R2 += rcnt[(size_t)r]; }
    S.all.reserve(4 * std::max<uint64_t>(R2, 1));
    {
        const void *snd[4] = {S.pad.p, S.pad.p + K, S.pad.p + 2 * K, S.pad.p + 3 * K};
        void *rcv[4] = {S.all.p, S.all.p + R2, S.all.p + 2 * R2, S.all.p + 3 * R2};
        C.all_to_all_dev(E, 4, snd, soff.data(), scnt.data(), rcv, roff.data(), rcnt.data(), &t_wait, &t_move);
    }
    E.stats.exchange_bytes += 16 * (R2 - rcnt[(size_t)me]);
    E.stats.phase_seconds[3] += t3.seconds() + t_part;
    E.stats.exchange2_seconds[0] += t_part; E.stats.exchange2_seconds[1] += t_counts;
    E.stats.exchange2_seconds[2] += t_wait; E.stats.exchange2_seconds[3] += t_move;
    uint64_t kept;
    {
        Turn turn(C, &E);
        Timer t4;
        kept = E.import_hits_dev(R2, (uint32_t *)S.all.p, (uint32_t *)S.all.p + R2, S.all.p + 2 * R2, S.all.p + 3 * R2, 0, 1);
        E.stats.phase_seconds[4] += t4.seconds();
    }
    E.stats.exchange_seconds += tm.seconds();
    return kept;
}

uint64_t cluster_step(Engine &E, Comm &C, int target_shards, uint32_t *assign) {
    if (!E.have_db) fail(UC_ERR_ARGS, "no database loaded");
    UC_HIP(hipSetDevice(E.device));
    const uint32_t n = E.hdb.n;
    {
        Turn turn(C, &E);
        Timer tp;
        prefilter_cell(E, C.world, target_shards, C.rank);
        E.stats.phase_seconds[0] += tp.seconds();
    }
    if (C.world > 1 || C.uses_rccl) (void)exchange_hits(E, C);
    const uint64_t aln_before = E.stats.n_gapped_alignments;
    {
        Turn turn(C, &E);
        Timer ta;
        E.align(0, n);
        E.stats.phase_seconds[5] += ta.seconds();
    }
    const uint64_t n_aln = E.stats.n_gapped_alignments - aln_before;   // pairs this rank aligned: its installed lists minus what an optional length gate (UC-1/L) ruled out
    if (C.world == 1 && !C.uses_rccl) {   // one rank: the graph is built straight from the device-resident edge list
        if (!assign && n) fail(UC_ERR_ARGS, "cluster_step: rank 0 needs an assignment buffer");
        Timer tc;
        const uint64_t ne = E.edges_on_host ? E.edges.size() / 2 : E.n_edges_dev;
        E.set_cover_own_edges(n, assign);
        E.stats.algorithmic_bytes[UC_ST_SETCOVER] += 8ull * ne + 4ull * n;
        E.stats.stage_seconds[UC_ST_SETCOVER] += tc.seconds();
        return n_aln;
    }
    Timer te;
    const uint32_t *dev_all = nullptr;
    const uint64_t n_all = C.gather_edges_dev(E, &dev_all);
    E.stats.exchange_seconds += C.world > 1 ? te.seconds() : 0.0;
    if (C.rank == 0) {
        if (!assign && n) fail(UC_ERR_ARGS, "cluster_step: rank 0 needs an assignment buffer");
        Timer tc;
        if (n_all < (1ull << 31)) E.set_cover_graph(n, nullptr, dev_all, n_all, assign);     // the graph is built from the buffer the edges landed in
        else {   // beyond the 32-bit positions of the device graph build: the all-host cover
            std::vector<uint32_t> all(2 * n_all);
            UC_HIP(hipMemcpy(all.data(), dev_all, 2 * n_all * 4, hipMemcpyDeviceToHost));
            set_cover(n, all.data(), n_all, assign);
        }
        E.stats.algorithmic_bytes[UC_ST_SETCOVER] += 8ull * n_all + 4ull * n;
        E.stats.stage_seconds[UC_ST_SETCOVER] += tc.seconds();
        E.stats.phase_seconds[7] += tc.seconds();
    }
    return n_aln;
}

}  // namespace uc
