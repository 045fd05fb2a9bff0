// uc_engine.cpp — host orchestration of the per-GPU engine: DB upload, batching of the gapped stage
// (E5/E6), kernel-level entry points.  Prefilter orchestration (E1-E4) lives in uc_prefilter.hip.
#include "uc_engine.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>

namespace uc {

// The class kernels of a pass are spread over 1 + N_AUX = 8 streams.  The HIP runtime maps streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4) when it initialises, so the library asks for 8 when it is loaded;
// a value the host process has set, or a runtime that is already initialised, is left alone (4 queues are correct,
// only less concurrent: the gapped stage of a 1/8 share took 105 ms instead of 88 ms).
namespace {
struct HwQueues {
    HwQueues() { setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0); }
} g_hw_queues;
}  // namespace

Engine::Engine(const Params &pp, int dev) : p(pp) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        fail(UC_ERR_DEVICE, "no HIP device available (%s); this engine has no CPU fallback",
             e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    if (dev < 0) { UC_HIP(hipGetDevice(&dev)); }
    if (dev >= ndev) fail(UC_ERR_DEVICE, "device %d requested but only %d visible", dev, ndev);
    device = dev;
    UC_HIP(hipSetDevice(device));
    // UC_STREAM_PRIORITY=high|low (read per engine; tools/overlap_probe.py only): the co-residency probe of r06 gives the prefilter engine's streams
    // priority over the gapped stage's to see whether the dispatcher then lets the two stages share the chip (profiles/r06/overlap_*.json: it does not)
    int prio = 0;
    bool have_prio = false;
    if (const char *sp = getenv("UC_STREAM_PRIORITY")) {
        int lo = 0, hi = 0;      // numerically lower = higher priority
        UC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        if (!strcmp(sp, "high")) { prio = hi; have_prio = true; }
        else if (!strcmp(sp, "low")) { prio = lo; have_prio = true; }
    }
    auto make_stream = [&](hipStream_t *st) {
        if (have_prio) UC_HIP(hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio));
        else UC_HIP(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    };
    make_stream(&stream);
    UC_HIP(hipEventCreate(&ev0));
    UC_HIP(hipEventCreate(&ev1));
    UC_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    for (int i = 0; i < N_AUX; i++) {      // (creating these on a helper thread beside the upload bought nothing: tools/cold_stamps.sh, r4)
        make_stream(&aux[i]);
        UC_HIP(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
    }
    if (const char *ns = getenv("UC_STREAMS")) n_streams = std::max(1, std::min(N_AUX + 1, atoi(ns)));
    d_S3.reserve(A * A);
    d_SA.reserve(A * A);
    UC_HIP(hipMemcpy(d_S3.p, p.S3, A * A, hipMemcpyHostToDevice));
    UC_HIP(hipMemcpy(d_SA.p, p.SA, A * A, hipMemcpyHostToDevice));
    memset(&stats, 0, sizeof stats);
}

Engine::~Engine() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (int i = 0; i < N_AUX; i++) {
        if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
        if (aux[i]) (void)hipStreamDestroy(aux[i]);
    }
    park_prefilter_scratch(pre, device);
    pre = nullptr;
    park_align_scratch(aln, device);
    aln = nullptr;
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (stream) (void)hipStreamDestroy(stream);
}

double Engine::timed_ms_begin() { UC_HIP(hipEventRecord(ev0, stream)); return 0; }
double Engine::timed_ms_end() {
    UC_HIP(hipEventRecord(ev1, stream));
    UC_HIP(hipEventSynchronize(ev1));
    float ms = 0;
    UC_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    return ms;
}

// padded device layout for the sequences of `hdb` (h_poff, h_len, buffers reserved); returns the padded size
uint64_t Engine::plan_db_layout() {
    const uint32_t n = hdb.n;
    h_poff.assign((size_t)n + 1, 0);
    h_len.resize(n);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n; i++) {
        h_poff[i] = (uint32_t)tot;
        h_len[i] = hdb.len(i);
        tot += ((uint64_t)h_len[i] + 16 + 15) & ~15ull;   // >= 16 pad bytes, 16-byte aligned starts
        if (tot >= (1ull << 32) - 64) fail(UC_ERR_ARGS, "database too large for 32-bit device offsets");
    }
    h_poff[n] = (uint32_t)tot;
    d_s3.reserve(tot + 64);
    d_sa.reserve(tot + 64);
    d_lt.reserve(tot + 64 + 16);
    d_off.reserve((size_t)n + 1);
    d_len.reserve(std::max<size_t>(n, 1));
    UC_HIP(hipMemcpyAsync(d_off.p, h_poff.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, stream));
    if (n) UC_HIP(hipMemcpyAsync(d_len.p, h_len.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
    return tot;
}

void Engine::finish_db_install(Timer &tm) {
    const uint32_t n = hdb.n;
    ddb.n = n;
    ddb.s3 = d_s3.p; ddb.sa = d_sa.p; ddb.lt = d_lt.p + 16; ddb.off = d_off.p; ddb.len = d_len.p;
    ddb.S3 = d_S3.p; ddb.SA = d_SA.p;
    ddb.bias = nullptr;
    if (p.comp_bias && n) {      // rule UC-1/B (optional): per-residue bias of the resident database, computed where the letters are
        d_bias.reserve((size_t)h_poff[n] + 64);
        UC_HIP(hipMemsetAsync(d_bias.p, 0, (size_t)h_poff[n] + 64, stream));
        launch_comp_bias(ddb, p.comp_bias_milli, d_bias.p, stream);
        UC_HIP(hipStreamSynchronize(stream));
        UC_HIP(hipGetLastError());
        ddb.bias = d_bias.p;
    }
    have_db = true;
    hit_cnt.assign(n, 0);
    hit_off.assign((size_t)n + 1, 0);
    n_hits = 0;
    alns_valid = false;
    clear_edges();
    max_len = 1;
    for (uint32_t i = 0; i < n; i++) max_len = std::max(max_len, h_len[i]);
    stats.n_seqs = n;
    stats.n_residues = hdb.residues();
    stats.algorithmic_bytes[UC_ST_LOAD] += 2 * hdb.residues() + 12ull * n;
    stats.stage_seconds[UC_ST_LOAD] += tm.seconds();
}

void Engine::upload_db(bool keep_raw) {
    PressureScope ps(*this, 2);
    Timer tm;
    UC_HIP(hipSetDevice(device));
    const uint32_t n = hdb.n;
    const uint64_t tot = plan_db_layout();
    // the raw tracks go up as they are (two contiguous copies); the padded layout - 16-byte aligned sequence starts, >= 16 pad
    // letters behind every sequence, the interleaved letter-pair stream of the gapped kernels - is laid out on the device
    // (building it on the host cost 55 ms of a 75 ms upload at 47 M residues).  keep_raw: the raw copy stays resident, so that
    // sub-databases (the representatives of a cascade round) can be laid out from it without the host (upload_sub_db)
    const uint64_t raw = hdb.residues();
    DevBuf<uint8_t> l3, la;
    DevBuf<uint64_t> loff;
    DevBuf<uint8_t> &r3 = keep_raw ? d_raw3 : l3, &ra = keep_raw ? d_rawa : la;
    DevBuf<uint64_t> &roff = keep_raw ? d_rawoff : loff;
    r3.reserve(std::max<uint64_t>(raw, 1)); ra.reserve(std::max<uint64_t>(raw, 1)); roff.reserve((size_t)n + 1);
    if (raw) {
        UC_HIP(hipMemcpyAsync(r3.p, hdb.s3.data(), raw, hipMemcpyHostToDevice, stream));
        UC_HIP(hipMemcpyAsync(ra.p, hdb.sa.data(), raw, hipMemcpyHostToDevice, stream));
    }
    UC_HIP(hipMemcpyAsync(roff.p, hdb.off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, stream));
    launch_db_pad(n, d_off.p, d_len.p, (const uint32_t *)nullptr, roff.p, r3.p, ra.p, tot + 64, d_s3.p, d_sa.p, d_lt.p, stream);
    UC_HIP(hipStreamSynchronize(stream));
    UC_HIP(hipGetLastError());
    raw_n = keep_raw ? n : 0;
    raw_resident = keep_raw;
    finish_db_install(tm);
}

// the sequences `cur` (ascending ids of the database uploaded with keep_raw) become the engine's database: offsets on the
// host (full_off = raw offsets of that database), the sequence data gathered on the device from the resident raw copy
void Engine::upload_sub_db(const std::vector<uint32_t> &cur, const std::vector<uint64_t> &full_off) {
    PressureScope ps(*this, 2);
    Timer tm;
    UC_HIP(hipSetDevice(device));
    if (!raw_resident || full_off.size() != (size_t)raw_n + 1) fail(UC_ERR_GENERIC, "upload_sub_db: no resident raw database");
    HostDb sub;
    sub.n = (uint32_t)cur.size();
    sub.off.resize((size_t)sub.n + 1);
    uint64_t t = 0;
    for (uint32_t i = 0; i < sub.n; i++) {
        if (cur[i] >= raw_n) fail(UC_ERR_GENERIC, "upload_sub_db: sequence id out of range");
        sub.off[i] = t;
        t += full_off[cur[i] + 1] - full_off[cur[i]];
    }
    sub.off[sub.n] = t;
    hdb = std::move(sub);                      // offsets only: nothing on the host reads the letters of a resident database
    const uint32_t n = hdb.n;
    const uint64_t tot = plan_db_layout();
    DevBuf<uint32_t> dcur;
    dcur.reserve(std::max<uint32_t>(n, 1));
    if (n) UC_HIP(hipMemcpyAsync(dcur.p, cur.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
    launch_db_pad(n, d_off.p, d_len.p, dcur.p, d_rawoff.p, d_raw3.p, d_rawa.p, tot + 64, d_s3.p, d_sa.p, d_lt.p, stream);
    UC_HIP(hipStreamSynchronize(stream));
    UC_HIP(hipGetLastError());
    finish_db_install(tm);
}

void Engine::drop_scratch() {
    UC_HIP(hipSetDevice(device));
    UC_HIP(hipStreamSynchronize(stream));
    // parked, not freed: the next virtual rank on this device takes the same buffers (one set per device), so no rank pays for
    // tens of GB of hipMalloc inside its timed phase
    if (pre) { park_prefilter_scratch(pre, device); pre = nullptr; }
    if (aln) { park_align_scratch(aln, device); aln = nullptr; last_align_hits = 0; }
}

bool Engine::relieve_pressure(int stage) {
    (void)hipStreamSynchronize(stream);
    size_t f0 = 0, f1 = 0, tot = 0;
    (void)hipMemGetInfo(&f0, &tot);
    bool freed = false;
    // a scratch set goes only when no frame of its own stage is open on this engine - the innermost scope decides what is asked for, every open one what is pinned
    if (stage != 1 && !stage_frames[1] && aln) { free_align_scratch(aln); aln = nullptr; last_align_hits = 0; freed = true; }
    if (stage == 1 && aln && release_tb_matrices(aln)) freed = true;      // the gapped stage's own traceback-byte buffer, when no batch loop is using it
    if (stage != 0 && !stage_frames[0] && pre) { free_prefilter_scratch(pre); pre = nullptr; freed = true; }
    if (PrefilterScratch *x = take_parked_prefilter_scratch(device)) { free_prefilter_scratch(x); freed = true; }
    if (AlignScratch *x = take_parked_align_scratch(device)) { free_align_scratch(x); freed = true; }
    (void)hipMemGetInfo(&f1, &tot);
    if (getenv("UC_TIMING") && g_verbosity < 2)
        fprintf(stderr, "unicore-cluster[timing]: device memory ran out in the %s; %s work buffers released (%.1f -> %.1f GiB free)\n",
                stage == 0 ? "prefilter" : stage == 1 ? "gapped stage" : "database upload", freed ? "the other stage's" : "no", (double)f0 / (1ull << 30), (double)f1 / (1ull << 30));
    logf(2, "unicore-cluster: device memory ran out in the %s; %s work buffers released (%.1f -> %.1f GiB free)\n",
         stage == 0 ? "prefilter" : stage == 1 ? "gapped stage" : "database upload", freed ? "the other stage's" : "no", (double)f0 / (1ull << 30), (double)f1 / (1ull << 30));
    return freed;
}

void Engine::ungapped_batch(uint64_t n, const uint32_t *q, const uint32_t *t, const int32_t *diag, int32_t *out) {
    if (!have_db) fail(UC_ERR_ARGS, "no database loaded");
    if (n == 0) return;
    UC_HIP(hipSetDevice(device));
    for (uint64_t i = 0; i < n; i++)
        if (q[i] >= hdb.n || t[i] >= hdb.n) fail(UC_ERR_ARGS, "ungapped_batch: sequence id out of range");
    DevBuf<uint32_t> dq, dt;
    DevBuf<int32_t> dd, ds;
    dq.reserve(n); dt.reserve(n); dd.reserve(n); ds.reserve(n);
    UC_HIP(hipMemcpyAsync(dq.p, q, n * 4, hipMemcpyHostToDevice, stream));
    UC_HIP(hipMemcpyAsync(dt.p, t, n * 4, hipMemcpyHostToDevice, stream));
    UC_HIP(hipMemcpyAsync(dd.p, diag, n * 4, hipMemcpyHostToDevice, stream));
    launch_ungapped(ddb, n, dq.p, dt.p, dd.p, ds.p, nullptr, stream);
    UC_HIP(hipGetLastError());
    UC_HIP(hipMemcpyAsync(out, ds.p, n * 4, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipStreamSynchronize(stream));
}

void Engine::set_hits(const uint32_t *counts, const uc_hit *h, bool check_max_seqs) {
    const uint32_t n = hdb.n;
    UC_HIP(hipSetDevice(device));
    hit_cnt.assign(counts, counts + n);
    hit_off.assign((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (check_max_seqs && counts[i] > (uint32_t)p.max_seqs) fail(UC_ERR_ARGS, "hit list of query %u longer than max_seqs", i);
        hit_off[i + 1] = hit_off[i] + counts[i];
    }
    n_hits = hit_off[n];
    std::vector<uint32_t> hq(n_hits), ht(n_hits);
    std::vector<int32_t> hs(n_hits), hd(n_hits);
    for (uint32_t q = 0; q < n; q++)
        for (uint64_t k = hit_off[q]; k < hit_off[q + 1]; k++) {
            if (h[k].target >= n) fail(UC_ERR_ARGS, "hit target out of range");
            hq[k] = q; ht[k] = h[k].target; hs[k] = h[k].score; hd[k] = h[k].diag;
        }
    const size_t cap = std::max<uint64_t>(n_hits, 1);
    d_hq.reserve(cap); d_ht.reserve(cap); d_hs.reserve(cap); d_hd.reserve(cap);
    if (n_hits) {
        UC_HIP(hipMemcpy(d_hq.p, hq.data(), n_hits * 4, hipMemcpyHostToDevice));
        UC_HIP(hipMemcpy(d_ht.p, ht.data(), n_hits * 4, hipMemcpyHostToDevice));
        UC_HIP(hipMemcpy(d_hs.p, hs.data(), n_hits * 4, hipMemcpyHostToDevice));
        UC_HIP(hipMemcpy(d_hd.p, hd.data(), n_hits * 4, hipMemcpyHostToDevice));
    }
    alns_valid = false;
    clear_edges();
}

void Engine::get_hits(uc_hit *out) const { get_hits_range(0, n_hits, out); }

void Engine::get_hits_range(uint64_t begin, uint64_t k, uc_hit *out) const {
    if (!k) return;
    if (begin + k > n_hits) fail(UC_ERR_ARGS, "get_hits_range: range outside the hit lists");
    UC_HIP(hipSetDevice(device));
    std::vector<uint32_t> ht(k);
    std::vector<int32_t> hs(k), hd(k);
    UC_HIP(hipMemcpy(ht.data(), d_ht.p + begin, k * 4, hipMemcpyDeviceToHost));
    UC_HIP(hipMemcpy(hs.data(), d_hs.p + begin, k * 4, hipMemcpyDeviceToHost));
    UC_HIP(hipMemcpy(hd.data(), d_hd.p + begin, k * 4, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < k; i++) { out[i].target = ht[i]; out[i].score = hs[i]; out[i].diag = hd[i]; }
}

void Engine::get_alns(uint64_t begin, uint64_t n, uc_aln *out) const {
    if (!n) return;
    if (!alns_valid) fail(UC_ERR_ARGS, "no alignment records: call align first");
    UC_HIP(hipSetDevice(device));
    UC_HIP(hipMemcpy(out, d_alns.p + begin, n * sizeof(uc_aln), hipMemcpyDeviceToHost));
}

// host merge of per-shard hit lists (multi-GPU exchange): per query keep the top
// max_seqs under the frozen order (score desc, target asc)
void merge_hits(uint32_t n, int max_seqs, int n_parts, const uint32_t *const *counts, const uc_hit *const *hits,
                std::vector<uint32_t> &out_cnt, std::vector<uc_hit> &out_hits) {
    // per-shard start of every query's list, merged size per query, then the per-query merges in parallel
    std::vector<std::vector<uint64_t>> pos((size_t)n_parts, std::vector<uint64_t>((size_t)n + 1, 0));
    out_cnt.assign(n, 0);
    std::vector<uint64_t> ooff((size_t)n + 1, 0);
    for (int s = 0; s < n_parts; s++)
        for (uint32_t q = 0; q < n; q++) pos[s][q + 1] = pos[s][q] + counts[s][q];
    for (uint32_t q = 0; q < n; q++) {
        uint64_t tot = 0;
        for (int s = 0; s < n_parts; s++) tot += counts[s][q];
        out_cnt[q] = (uint32_t)std::min<uint64_t>(tot, (uint64_t)max_seqs);
        ooff[q + 1] = ooff[q] + out_cnt[q];
    }
    out_hits.resize(ooff[n]);
    const unsigned nthr = n < 4096 ? 1u : std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::string> errors(nthr);
    auto work = [&](unsigned t) {
        std::vector<uc_hit> tmp;
        for (uint32_t q = (uint32_t)((uint64_t)n * t / nthr); q < (uint32_t)((uint64_t)n * (t + 1) / nthr); q++) {
            tmp.clear();
            for (int s = 0; s < n_parts; s++) tmp.insert(tmp.end(), hits[s] + pos[s][q], hits[s] + pos[s][q + 1]);
            std::sort(tmp.begin(), tmp.end(), [](const uc_hit &a, const uc_hit &b) {
                return a.score != b.score ? a.score > b.score : a.target < b.target;
            });
            for (size_t i = 1; i < tmp.size(); i++)
                if (tmp[i].target == tmp[i - 1].target && tmp[i].score == tmp[i - 1].score && errors[t].empty())
                    errors[t] = "merge_hits: target " + std::to_string(tmp[i].target) + " appears in two shards for query " + std::to_string(q);
            std::copy(tmp.begin(), tmp.begin() + out_cnt[q], out_hits.begin() + ooff[q]);
        }
    };
    if (nthr == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthr; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    for (const std::string &e : errors)
        if (!e.empty()) fail(UC_ERR_ARGS, "%s", e.c_str());
}

}  // namespace uc
