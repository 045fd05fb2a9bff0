// uc_multi.h — the multi-GPU layout of the cluster path inside the library: one Engine per GPU, one host
// thread per engine (uc_cluster, num_gpus > 1) or one process per engine (uc_comm_* of the C ABI, bench.py under
// torch.distributed.run), the per-shard hit lists all-gathered device to device with RCCL over xGMI.
//
//   rank r : index target shard r % T -> match query group r / T against it (E1-E4)       [Q x T = world]
//   exchange 1: the shard lists go to the query's HOME rank (ragged all-to-all: grouped ncclSend / ncclRecv over xGMI), which
//               merges its 1/N of the queries under (score desc, target asc) and keeps the top max_seqs
//   exchange 2: the surviving pairs go to the rank that OWNS them (hash of the unordered pair's representative query, so mutual
//               hits share their DP on one rank) -> E5/E6 on its share
//   rank 0 : accepted edges of all ranks -> host set cover (E7)
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

#include "uc_engine.h"

struct ncclComm;   // rccl.h

namespace uc {

// Ranks that live in ONE process (host threads) meet here: a reusable barrier that also carries a failure flag — a rank
// that throws marks the group failed, and every other rank leaves its next barrier with an error instead of waiting
// forever (or walking into a collective its peer will never join).
struct LocalGroup {
    explicit LocalGroup(int world) : world(world), ptr((size_t)world, nullptr), val((size_t)world, 0) {}
    const int world;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    bool failed = false;
    std::vector<const void *> ptr;   // per-rank published pointer (device buffer or host vector)
    std::vector<uint64_t> val;       // per-rank published value
    // emulation aid for virtual ranks (several engines on ONE device): with `serialize` the compute phases of the ranks take
    // turns on the GPU, so every rank's phase times are those of a rank with a GPU of its own (UC_VIRTUAL_SERIAL=1)
    bool serialize = false;
    std::mutex turn;
    void barrier();                  // throws Error(UC_ERR_GENERIC) if the group failed
    void fail_all();                 // called by a rank on its way out with an exception
};

struct CommScratch;   // device staging buffers of the exchange (uc_multi.cpp)

struct Comm {
    int rank = 0, world = 1;
    LocalGroup *grp = nullptr;       // set when all ranks are threads of this process
    // set when the ranks sit on distinct devices (RCCL); null = in-process copies (virtual GPUs of the tests).  Atomic: abort() takes the handle
    // from any thread WITHOUT waiting for an enqueue in flight (an enqueue can block inside ncclGroupEnd on a peer — the very call an abort
    // has to interrupt; ncclCommAbort is the one RCCL call that may run beside a blocked call on the same communicator)
    std::atomic<ncclComm *> nccl{nullptr};
    std::unique_ptr<CommScratch> scratch;
    std::atomic<bool> aborted{false};
    std::atomic<int> in_flight{0};   // threads inside an RCCL call on `nccl` (nccl_enqueue); abort() lets them leave before it reclaims the handle
    bool uses_rccl = false;          // the communicator was created over RCCL (stays true after an abort took the handle)
    Comm();
    ~Comm();
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;

    // failure path: ncclCommAbort, so that a peer blocked inside a collective of this communicator returns (its next barrier
    // then throws); the handle is gone afterwards.  Safe to call from any thread, once or more.  Takes no lock; the reclamation of the handle
    // is DEFERRED until the threads that entered an RCCL call with it have left (in_flight == 0) - bounded: after a grace period an enqueue
    // that is still inside is blocked on the dying peer, and ncclCommAbort is the only call that gets it out.
    void abort();
    void mark_aborted();             // first half of abort(): new enqueues fail from here on (abort_all marks every communicator before it waits on any)
    void barrier(Engine &E);
    void all_gather_u64(Engine &E, uint64_t v, uint64_t *out /* world */);
    void all_gather_u64s(Engine &E, const uint64_t *v, int k, uint64_t *out /* world x k */);
    // ragged all-to-all of `na` parallel arrays of 4-byte elements (offsets and counts in elements, per peer)
    // t_wait / t_move (nullable) accumulate the seconds spent waiting for the peers' rendezvous and in the data movement itself
    void all_to_all_dev(Engine &E, int na, const void *const *send, const uint64_t *send_off, const uint64_t *send_cnt,
                        void *const *recv, const uint64_t *recv_off, const uint64_t *recv_cnt, double *t_wait = nullptr, double *t_move = nullptr);
    // recv holds world x bytes; send/recv are device buffers of E's device
    void all_gather_dev(Engine &E, const void *send, void *recv, size_t bytes);
    void broadcast_dev(Engine &E, void *buf, size_t bytes, int root);
    // accepted edges of every rank, concatenated in rank order, on rank 0 — device to device: every rank's device-resident edge list goes straight into ONE device buffer of rank 0 (grouped ncclSend / ncclRecv over
    // xGMI; virtual ranks: device copies) — no D2H on the senders, no host concatenation, no H2D on rank 0, whose graph build reads the buffer where it
    // lands.  Returns the total number of edges (pairs) over all ranks on EVERY rank; *dev_out (rank 0 only) stays valid until the next gather.
    uint64_t gather_edges_dev(Engine &E, const uint32_t **dev_out);
};

// emulation aid (UC_VIRTUAL_SERIAL=1, virtual ranks only): the compute phases of the ranks take turns on the one physical GPU,
// so that a rank's phase_seconds are what it would measure with a GPU of its own (tools/critical_path.py)
struct Turn {
    LocalGroup *g;
    Engine *e = nullptr;
    explicit Turn(Comm &C, Engine *E = nullptr) : g(C.grp && C.grp->serialize ? C.grp : nullptr), e(E) { if (g) g->turn.lock(); }
    // eight engines' work buffers do not fit one GPU at BASELINE configs[2] size: a rank gives its scratch back at the end of its turn
    // (a destructor must not throw: a HIP error while handing buffers back is dropped here and surfaces at the rank's next HIP call)
    ~Turn() { if (g) { if (e) { try { e->drop_scratch(); } catch (...) {} } g->turn.unlock(); } }
    Turn(const Turn &) = delete;
    Turn &operator=(const Turn &) = delete;
};

// Q x T grid: rank r indexes target shard r % T and matches query group r / T.  target_shards = 0 means T = world — the
// north-star layout ("target DB range-partitioned across the GPUs"); it must divide world.
struct GridCell { uint32_t tb, te, qb, qe; };
std::vector<std::pair<uint32_t, uint32_t>> shard_ranges(const std::vector<uint32_t> &len, int parts);
void grid_shape(int world, int target_shards, int *Q, int *T);
GridCell grid_cell(const std::vector<uint32_t> &len, int world, int target_shards, int rank);

// E1-E4 of this rank's share of the pass.  T = N under a symmetric matrix (the north-star layout): the shard x shard grid needs only half of its
// off-diagonal blocks, because the hits of (t, q) are those of (q, t) with the diagonal negated — rank r matches its target shard against its own
// sequences and against the queries of the next N/2 shards in cyclic order (the shard at distance exactly N/2 only from the lower half of the
// ranks), and every pair of those blocks also yields the candidate of the pair the other way round, which travels to its query's home rank with
// exchange 1 like any other record: 5 (ranks 0-3) or 4 blocks per rank instead of 8 at N = 8.  Q x T grids with Q > 1, non-symmetric matrices and
// UC_PREFILTER_SYMMETRIC=0 match the rank's whole cell as before.
void prefilter_cell(Engine &E, int world, int target_shards, int rank);

// all-gather + merge + ownership filter of the engine's hit lists; returns the pairs this rank now owns
uint64_t exchange_hits(Engine &E, Comm &C);
// one pass of the sharded path on this rank (prefilter of the rank's cell -> exchange -> E5/E6 -> edges to rank 0 -> set
// cover on rank 0).  assign (n entries) is written on rank 0 only.  Returns the gapped alignments of this rank.
uint64_t cluster_step(Engine &E, Comm &C, int target_shards, uint32_t *assign);

// RCCL plumbing for the C ABI
void comm_unique_id(uint8_t id[128]);
void comm_info(const Comm &C, int *count, int *rank, int *device);   // asks RCCL
void comm_init_rank(Comm &C, const uint8_t id[128], int rank, int world, int device);
// all communicators of an in-process group from ONE thread (ncclCommInitAll): either every rank gets its handle or the call
// fails as a whole — no rank can be left waiting inside a collective init for a peer that died while building its engine
void comm_init_all(const std::vector<Comm *> &comms, const std::vector<int> &devices);
void abort_all(const std::vector<Comm *> &comms);     // failure path: mark every communicator, then the bounded waits + aborts (grace periods overlap)

}  // namespace uc
