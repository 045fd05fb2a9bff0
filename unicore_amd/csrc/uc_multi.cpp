// uc_multi.cpp — multi-GPU exchange of the cluster path: grid plan, RCCL all-gather of the per-shard hit
// lists, device merge, edge gather.  See uc_multi.h for the layout.  RCCL is called from here — the library itself —
// so `unicore cluster` / `foldseek cluster` use every GPU of the node with no Python in the process.
#include "uc_multi.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

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
    if (nccl) (void)ncclCommDestroy(nccl);
}

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
    UC_NCCL(ncclCommInitRank(&C.nccl, world, u, rank));
}

// A communicator that owns an RCCL handle always goes through RCCL, even with one rank: a 1-rank communicator on the
// single-GPU test box exercises exactly the calls an 8-rank run makes.
void Comm::barrier(Engine &E) {
    if (world == 1 && !nccl) return;
    if (grp) { grp->barrier(); return; }
    std::vector<uint64_t> dummy((size_t)world);
    all_gather_u64(E, 0, dummy.data());   // cross-process: the smallest collective doubles as the barrier
}

void Comm::all_gather_u64(Engine &E, uint64_t v, uint64_t *out) {
    if (world == 1 && !nccl) { out[0] = v; return; }
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
    UC_NCCL(ncclAllGather(S.sz_send.p, S.sz_recv.p, 1, ncclUint64, nccl, E.stream));
    UC_HIP(hipMemcpyAsync(out, S.sz_recv.p, (size_t)world * 8, hipMemcpyDeviceToHost, E.stream));
    UC_HIP(hipStreamSynchronize(E.stream));
}

void Comm::all_gather_dev(Engine &E, const void *send, void *recv, size_t bytes) {
    if (world == 1 && !nccl) {
        if (bytes) UC_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, E.stream));
        UC_HIP(hipStreamSynchronize(E.stream));
        return;
    }
    if (nccl) {
        if (grp) grp->barrier();   // a rank that failed earlier must not leave its peers inside the collective
        UC_NCCL(ncclAllGather(send, recv, bytes, ncclUint8, nccl, E.stream));
        UC_HIP(hipStreamSynchronize(E.stream));
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
    if (world == 1 && !nccl) return;
    if (nccl) {
        if (grp) grp->barrier();
        UC_NCCL(ncclBroadcast(buf, buf, bytes, ncclUint8, root, nccl, E.stream));
        UC_HIP(hipStreamSynchronize(E.stream));
        return;
    }
    UC_HIP(hipStreamSynchronize(E.stream));
    if (rank == root) grp->ptr[(size_t)root] = buf;
    grp->barrier();
    if (rank != root && bytes) UC_HIP(hipMemcpyAsync(buf, grp->ptr[(size_t)root], bytes, hipMemcpyDefault, E.stream));
    UC_HIP(hipStreamSynchronize(E.stream));
    grp->barrier();
}

void Comm::gather_edges(Engine &E, std::vector<uint32_t> &out) {
    out.clear();
    E.host_edges();                     // the exchange below works on the host copies
    if (world == 1 && !nccl) { out = E.edges; return; }
    if (grp) {   // threads of one process: rank 0 reads its peers' host vectors ("edges are copied D2H per GPU ... the host runs set cover once")
        grp->ptr[(size_t)rank] = &E.edges;
        grp->barrier();
        if (rank == 0) {
            size_t tot = 0;
            for (int r = 0; r < world; r++) tot += ((const std::vector<uint32_t> *)grp->ptr[(size_t)r])->size();
            out.reserve(tot);
            for (int r = 0; r < world; r++) {
                const std::vector<uint32_t> &v = *(const std::vector<uint32_t> *)grp->ptr[(size_t)r];
                out.insert(out.end(), v.begin(), v.end());
            }
        }
        grp->barrier();
        return;
    }
    // one process per GPU: sizes, then every rank sends its list to rank 0 (grouped point-to-point over xGMI)
    std::vector<uint64_t> sz((size_t)world);
    all_gather_u64(E, E.edges.size(), sz.data());
    CommScratch &S = *scratch;
    const size_t mine = E.edges.size();
    S.e_send.reserve(std::max<size_t>(mine, 1));
    if (mine) UC_HIP(hipMemcpyAsync(S.e_send.p, E.edges.data(), mine * 4, hipMemcpyHostToDevice, E.stream));
    size_t tot = 0;
    for (uint64_t s : sz) tot += s;
    if (rank == 0) S.e_recv.reserve(std::max<size_t>(tot, 1));
    UC_NCCL(ncclGroupStart());
    if (rank == 0) {
        size_t o = sz[0];
        for (int r = 1; r < world; r++) {
            if (sz[(size_t)r]) UC_NCCL(ncclRecv(S.e_recv.p + o, sz[(size_t)r], ncclUint32, r, nccl, E.stream));
            o += sz[(size_t)r];
        }
    } else if (mine) {
        UC_NCCL(ncclSend(S.e_send.p, mine, ncclUint32, 0, nccl, E.stream));
    }
    UC_NCCL(ncclGroupEnd());
    if (rank == 0) {
        out.resize(tot);
        if (mine) memcpy(out.data(), E.edges.data(), mine * 4);
        if (tot > mine) UC_HIP(hipMemcpyAsync(out.data() + mine, S.e_recv.p + mine, (tot - mine) * 4, hipMemcpyDeviceToHost, E.stream));
    }
    UC_HIP(hipStreamSynchronize(E.stream));
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

// ---------------------------------------------------------------------------------------------- the exchange
namespace {
uint64_t one_shot_limit() {   // records in the union above which the lists are merged shard by shard (peak = two lists, not world)
    if (const char *e = getenv("UC_EXCHANGE_LIMIT")) return strtoull(e, nullptr, 10);
    return 256ull << 20;
}
}  // namespace

uint64_t exchange_hits(Engine &E, Comm &C) {
    Timer tm;
    UC_HIP(hipSetDevice(E.device));
    const int W = C.world;
    CommScratch &S = *C.scratch;
    const uint64_t nloc = E.n_hits;
    std::vector<uint64_t> sizes((size_t)W);
    C.all_gather_u64(E, nloc, sizes.data());
    uint64_t total = 0, m = 1;
    for (uint64_t s : sizes) { total += s; m = std::max(m, s); }
    uint64_t kept = 0;
    if (total > one_shot_limit()) {
        // BASELINE configs[2] scale: one broadcast per rank, top-M truncation after every merge
        S.own.reserve(4 * std::max<uint64_t>(nloc, 1));
        int32_t *own = S.own.p;
        if (nloc) E.export_hits_dev((uint32_t *)own, (uint32_t *)own + nloc, own + 2 * nloc, own + 3 * nloc);
        uint64_t nacc = 0;
        for (int r = 0; r < W; r++) {
            const uint64_t nr = sizes[(size_t)r];
            if (!nr) continue;
            S.part.reserve(4 * nr);
            if (r == C.rank) UC_HIP(hipMemcpyAsync(S.part.p, own, 16 * nr, hipMemcpyDeviceToDevice, E.stream));
            C.broadcast_dev(E, S.part.p, 16 * nr, r);
            E.stats.exchange_bytes += 16 * nr;
            if (!nacc) {
                S.acc.reserve(4 * nr);
                UC_HIP(hipMemcpyAsync(S.acc.p, S.part.p, 16 * nr, hipMemcpyDeviceToDevice, E.stream));
                UC_HIP(hipStreamSynchronize(E.stream));
                nacc = nr;
                continue;
            }
            const uint64_t tot = nacc + nr;
            S.cat.reserve(4 * tot);
            for (int k = 0; k < 4; k++) {
                UC_HIP(hipMemcpyAsync(S.cat.p + k * tot, S.acc.p + k * nacc, 4 * nacc, hipMemcpyDeviceToDevice, E.stream));
                UC_HIP(hipMemcpyAsync(S.cat.p + k * tot + nacc, S.part.p + k * nr, 4 * nr, hipMemcpyDeviceToDevice, E.stream));
            }
            UC_HIP(hipStreamSynchronize(E.stream));
            nacc = E.import_hits_dev(tot, (uint32_t *)S.cat.p, (uint32_t *)S.cat.p + tot, S.cat.p + 2 * tot, S.cat.p + 3 * tot, 0, 1);   // merge + top-M, no ownership yet
            S.acc.reserve(4 * std::max<uint64_t>(nacc, 1));
            if (nacc) E.export_hits_dev((uint32_t *)S.acc.p, (uint32_t *)S.acc.p + nacc, S.acc.p + 2 * nacc, S.acc.p + 3 * nacc);
        }
        kept = E.import_hits_dev(nacc, (uint32_t *)S.acc.p, (uint32_t *)S.acc.p + nacc, S.acc.p + 2 * nacc, S.acc.p + 3 * nacc, (uint32_t)C.rank, (uint32_t)W);
    } else {
        // one padded all-gather: [4][m] int32 per rank -> [W][4][m]
        S.pad.reserve(4 * m);
        S.all.reserve(4 * m * (uint64_t)W);
        UC_HIP(hipMemsetAsync(S.pad.p, 0, 16 * m, E.stream));
        if (nloc) E.export_hits_dev((uint32_t *)S.pad.p, (uint32_t *)S.pad.p + m, S.pad.p + 2 * m, S.pad.p + 3 * m);
        C.all_gather_dev(E, S.pad.p, S.all.p, 16 * m);
        E.stats.exchange_bytes += 16 * m * (uint64_t)W;
        S.cat.reserve(4 * std::max<uint64_t>(total, 1));
        uint64_t o = 0;
        for (int r = 0; r < W; r++) {
            const uint64_t nr = sizes[(size_t)r];
            for (int k = 0; k < 4 && nr; k++)
                UC_HIP(hipMemcpyAsync(S.cat.p + k * total + o, S.all.p + ((uint64_t)r * 4 + k) * m, 4 * nr, hipMemcpyDeviceToDevice, E.stream));
            o += nr;
        }
        UC_HIP(hipStreamSynchronize(E.stream));
        kept = E.import_hits_dev(total, (uint32_t *)S.cat.p, (uint32_t *)S.cat.p + total, S.cat.p + 2 * total, S.cat.p + 3 * total, (uint32_t)C.rank, (uint32_t)W);
    }
    E.stats.exchange_seconds += tm.seconds();
    return kept;
}

uint64_t cluster_step(Engine &E, Comm &C, int target_shards, uint32_t *assign) {
    if (!E.have_db) fail(UC_ERR_ARGS, "no database loaded");
    UC_HIP(hipSetDevice(E.device));
    const uint32_t n = E.hdb.n;
    const GridCell g = grid_cell(E.h_len, C.world, target_shards, C.rank);
    E.prefilter(g.tb, g.te, g.qb, g.qe);
    uint64_t n_aln = E.n_hits;
    if (C.world > 1 || C.nccl) n_aln = exchange_hits(E, C);
    E.align(0, n);
    if (C.world == 1 && !C.nccl) {   // one rank: the graph is built straight from the device-resident edge list
        if (!assign && n) fail(UC_ERR_ARGS, "cluster_step: rank 0 needs an assignment buffer");
        Timer tc;
        const uint64_t ne = E.edges_on_host ? E.edges.size() / 2 : E.n_edges_dev;
        E.set_cover_own_edges(n, assign);
        E.stats.algorithmic_bytes[UC_ST_SETCOVER] += 8ull * ne + 4ull * n;
        E.stats.stage_seconds[UC_ST_SETCOVER] += tc.seconds();
        return n_aln;
    }
    std::vector<uint32_t> all;
    Timer te;
    C.gather_edges(E, all);
    E.stats.exchange_seconds += C.world > 1 ? te.seconds() : 0.0;
    if (C.rank == 0) {
        if (!assign && n) fail(UC_ERR_ARGS, "cluster_step: rank 0 needs an assignment buffer");
        Timer tc;
        E.set_cover_device(n, all.data(), all.size() / 2, assign);
        E.stats.algorithmic_bytes[UC_ST_SETCOVER] += 8ull * (all.size() / 2) + 4ull * n;
        E.stats.stage_seconds[UC_ST_SETCOVER] += tc.seconds();
    }
    return n_aln;
}

}  // namespace uc
