// uc_multi.h — the multi-GPU layout of the cluster path inside the library: one Engine per GPU, one host
// thread per engine (uc_cluster, num_gpus > 1) or one process per engine (uc_comm_* of the C ABI, bench.py under
// torch.distributed.run), the per-shard hit lists all-gathered device to device with RCCL over xGMI.
//
//   rank r : index target shard r % T -> match query group r / T against it (E1-E4)       [Q x T = world]
//   exchange: ncclAllGather of the padded [4][m] int32 hit tensors (the ONE collective of the path)
//   every rank: device merge per query under (score desc, target asc), top max_seqs, keep the pairs it owns
//               (hash of the unordered pair's representative query) -> E5/E6 on its share
//   rank 0 : accepted edges of all ranks -> host set cover (E7)
#pragma once
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
    void barrier();                  // throws Error(UC_ERR_GENERIC) if the group failed
    void fail_all();                 // called by a rank on its way out with an exception
};

struct CommScratch;   // device staging buffers of the exchange (uc_multi.cpp)

struct Comm {
    int rank = 0, world = 1;
    LocalGroup *grp = nullptr;       // set when all ranks are threads of this process
    ncclComm *nccl = nullptr;        // set when the ranks sit on distinct devices (RCCL); null = in-process copies (virtual GPUs of the tests)
    std::unique_ptr<CommScratch> scratch;
    Comm();
    ~Comm();
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;

    void barrier(Engine &E);
    void all_gather_u64(Engine &E, uint64_t v, uint64_t *out /* world */);
    // recv holds world x bytes; send/recv are device buffers of E's device
    void all_gather_dev(Engine &E, const void *send, void *recv, size_t bytes);
    void broadcast_dev(Engine &E, void *buf, size_t bytes, int root);
    // accepted edges of every rank, concatenated in rank order, on rank 0 (empty elsewhere)
    void gather_edges(Engine &E, std::vector<uint32_t> &out);
};

// Q x T grid: rank r indexes target shard r % T and matches query group r / T.  target_shards = 0 means T = world — the
// north-star layout ("target DB range-partitioned across the GPUs"); it must divide world.
struct GridCell { uint32_t tb, te, qb, qe; };
std::vector<std::pair<uint32_t, uint32_t>> shard_ranges(const std::vector<uint32_t> &len, int parts);
void grid_shape(int world, int target_shards, int *Q, int *T);
GridCell grid_cell(const std::vector<uint32_t> &len, int world, int target_shards, int rank);

// all-gather + merge + ownership filter of the engine's hit lists; returns the pairs this rank now owns
uint64_t exchange_hits(Engine &E, Comm &C);
// one pass of the sharded path on this rank (prefilter of the rank's cell -> exchange -> E5/E6 -> edges to rank 0 -> set
// cover on rank 0).  assign (n entries) is written on rank 0 only.  Returns the gapped alignments of this rank.
uint64_t cluster_step(Engine &E, Comm &C, int target_shards, uint32_t *assign);

// RCCL plumbing for the C ABI
void comm_unique_id(uint8_t id[128]);
void comm_init_rank(Comm &C, const uint8_t id[128], int rank, int world, int device);

}  // namespace uc
