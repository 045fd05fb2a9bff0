// uc_capi.cpp — extern "C" boundary (include/unicore_cluster.h).  No exception crosses it: every entry
// point converts failures into the status classes of the header and records uc_last_error().
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <memory>
#include <mutex>
#include <thread>

#include "uc_engine.h"
#include "uc_multi.h"
#include "uc_t5.h"

using namespace uc;

struct uc_engine {
    std::unique_ptr<Engine> e;
};
struct uc_comm {
    Comm c;
};
struct uc_t5 {
    T5Model m;
};

namespace {

template <typename F>
int guard(F &&f) {
    try {
        f();
        return UC_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc &) {
        set_last_error("out of host memory");
        return UC_ERR_GENERIC;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return UC_ERR_GENERIC;
    }
}

Params params_from(const uc_opts *o, bool search = false) {
    Params p;
    if (search) { p.evalue = 10.0; p.max_seqs = 1000; p.want_tb = 1; }   // `foldseek search` defaults (EXT-UNVERIFIED)
    if (o) {
        if (o->struct_size != sizeof(uc_opts)) fail(UC_ERR_ARGS, "uc_opts.struct_size mismatch (%u != %zu)", o->struct_size, sizeof(uc_opts));
        p.threads = o->threads > 0 ? o->threads : 1;
        p.verbosity = o->verbosity;
        p.num_gpus = o->num_gpus;
        if (o->cluster_options) parse_cluster_options(o->cluster_options, p);
    }
    g_verbosity = p.verbosity;
    finalize_params(p, o && o->data_dir ? o->data_dir : "");
    return p;
}

void mkdir_p(const std::string &path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && cur != "/") {
                struct stat st;
                if (stat(cur.c_str(), &st) != 0 && mkdir(cur.c_str(), 0777) != 0) fail(UC_ERR_IO, "cannot create directory %s", cur.c_str());
            }
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
}

// workflow observer (uc_set_round_hook): read once per round by rank 0's thread
std::mutex g_hook_mu;
uc_round_hook g_round_hook = nullptr;
void *g_round_hook_user = nullptr;

void call_round_hook(Engine &E, int round, const std::vector<uint32_t> &ids) {
    uc_round_hook h; void *u;
    { std::lock_guard<std::mutex> lk(g_hook_mu); h = g_round_hook; u = g_round_hook_user; }
    if (!h) return;
    uc_engine view;                       // a non-owning view of the round's engine for the duration of the call
    view.e.reset(&E);
    struct Release { uc_engine &v; ~Release() { (void)v.e.release(); } } rel{view};
    h(u, round, E.hdb.n, ids.data(), E.p.kmer_thr, &view);
}

void require(const void *p, const char *what) {
    if (!p) fail(UC_ERR_ARGS, "%s must not be NULL", what);
}

}  // namespace

extern "C" {

const char *uc_last_error(void) { return last_error_cstr(); }
const char *uc_version(void) {
    // says which 3Di matrix a run without --mat3di would use: results with the stand-in are not Foldseek's
    static thread_local std::string v;
    struct stat st;
    const bool real = stat((default_data_dir() + "/mat3di.out").c_str(), &st) == 0;
    v = std::string("unicore-cluster-mi355x 0.2.0 (spec UC-1.1, gfx950; default workflow: linclust pre-step + 3-step cascade; 3Di matrix: ") +
        (real ? "data/mat3di.out" : "MISSING - only the synthetic stand-in is shipped, runs need UC_ALLOW_SYNTHETIC=1 or --mat3di") + ")";
    return v.c_str();
}

uint32_t uc_abi_version(void) { return UC_ABI_VERSION; }
size_t uc_stats_size(void) { return sizeof(uc_stats); }
void uc_set_round_hook(uc_round_hook hook, void *user) {
    std::lock_guard<std::mutex> lk(g_hook_mu);
    g_round_hook = hook; g_round_hook_user = user;
}

int uc_option_arity(const char *flag) { return flag ? option_arity(flag) : -1; }

void uc_release_scratch(void) {
    for (int d = 0; d < 16; d++) {
        free_prefilter_scratch(take_parked_prefilter_scratch(d));
        free_align_scratch(take_parked_align_scratch(d));
    }
}

int uc_check_options(const char *cluster_options) {
    return guard([&] { Params p; parse_cluster_options(cluster_options ? cluster_options : "", p); });
}

int uc_engine_create(const uc_opts *o, uc_engine **out) {
    return guard([&] {
        require(out, "out");
        *out = nullptr;
        Params p = params_from(o);
        auto h = std::make_unique<uc_engine>();
        h->e = std::make_unique<Engine>(p, o ? o->device : -1);
        *out = h.release();
    });
}

void uc_engine_destroy(uc_engine *e) { delete e; }

int uc_engine_load_db(uc_engine *e, const char *db_prefix) {
    return guard([&] {
        require(e, "engine"); require(db_prefix, "db_prefix");
        Timer tm;
        read_seq_db(db_prefix, e->e->hdb, true);
        e->e->stats.stage_seconds[UC_ST_LOAD] += tm.seconds();
        e->e->upload_db();
    });
}

int uc_engine_set_db(uc_engine *e, uint32_t n, const uint64_t *off, const uint8_t *s3, const uint8_t *sa) {
    return guard([&] {
        require(e, "engine"); require(off, "off");
        if (n >= (1u << 24)) fail(UC_ERR_ARGS, "this build supports < 2^24 sequences");
        HostDb &d = e->e->hdb;
        d = HostDb();
        d.n = n;
        d.off.assign(off, off + n + 1);
        if (d.off[0] != 0) fail(UC_ERR_ARGS, "off[0] must be 0");
        for (uint32_t i = 0; i < n; i++) {
            if (d.off[i + 1] < d.off[i]) fail(UC_ERR_ARGS, "offsets must be non-decreasing");
            if (d.off[i + 1] - d.off[i] > 65535) fail(UC_ERR_ARGS, "sequence %u longer than 65535", i);
        }
        const uint64_t tot = d.off[n];
        if (tot) { require(s3, "s3"); require(sa, "sa"); }
        d.s3.assign(s3, s3 + tot);
        d.sa.assign(sa, sa + tot);
        for (uint64_t k = 0; k < tot; k++)
            if (d.s3[k] > 20 || d.sa[k] > 20) fail(UC_ERR_ARGS, "sequence codes must be in 0..20");
        d.keys.resize(n);
        for (uint32_t i = 0; i < n; i++) d.keys[i] = i;
        e->e->upload_db();
    });
}

uint32_t uc_engine_num_seqs(const uc_engine *e) { return e && e->e ? e->e->hdb.n : 0; }

int uc_engine_prefilter(uc_engine *e, uint32_t tbegin, uint32_t tend) {
    return guard([&] { require(e, "engine"); e->e->prefilter(tbegin, tend); });
}

int uc_engine_prefilter_range(uc_engine *e, uint32_t tbegin, uint32_t tend, uint32_t qbegin, uint32_t qend) {
    return guard([&] { require(e, "engine"); e->e->prefilter(tbegin, tend, qbegin, qend); });
}

int uc_engine_hits_size(const uc_engine *e, uint64_t *n_hits) {
    return guard([&] { require(e, "engine"); require(n_hits, "n_hits"); *n_hits = e->e->n_hits; });
}

int uc_engine_hits_get(const uc_engine *e, uint32_t *counts, uc_hit *hits) {
    return guard([&] {
        require(e, "engine");
        const Engine &E = *e->e;
        if (counts && E.hdb.n) memcpy(counts, E.hit_cnt.data(), (size_t)E.hdb.n * 4);
        if (hits) E.get_hits(hits);
    });
}

int uc_engine_hits_get_range(const uc_engine *e, uint32_t qbegin, uint32_t qend, uint32_t *counts, uc_hit *hits) {
    return guard([&] {
        require(e, "engine");
        const Engine &E = *e->e;
        if (!E.have_db) fail(UC_ERR_ARGS, "hits_get_range: no database loaded");
        if (qbegin > qend || qend > E.hdb.n) fail(UC_ERR_ARGS, "hits_get_range: bad query range");
        if (counts && qend > qbegin) memcpy(counts, E.hit_cnt.data() + qbegin, (size_t)(qend - qbegin) * 4);
        const uint64_t b = E.hit_off[qbegin], k = E.hit_off[qend] - b;
        if (hits && k) E.get_hits_range(b, k, hits);
    });
}

int uc_hits_merge(uint32_t n_seqs, int32_t max_seqs, int n_parts, const uint32_t *const *counts, const uc_hit *const *hits,
                  uint32_t *out_counts, uc_hit *out_hits, uint64_t out_capacity, uint64_t *out_n) {
    return guard([&] {
        require(counts, "counts"); require(hits, "hits"); require(out_counts, "out_counts"); require(out_n, "out_n");
        if (n_parts < 1 || max_seqs < 1) fail(UC_ERR_ARGS, "n_parts and max_seqs must be >= 1");
        std::vector<uint32_t> c;
        std::vector<uc_hit> h;
        merge_hits(n_seqs, max_seqs, n_parts, counts, hits, c, h);
        *out_n = h.size();
        if (h.size() > out_capacity) fail(UC_ERR_ARGS, "uc_hits_merge: output capacity %llu < %zu", (unsigned long long)out_capacity, h.size());
        if (n_seqs) memcpy(out_counts, c.data(), (size_t)n_seqs * 4);
        if (!h.empty()) { require(out_hits, "out_hits"); memcpy(out_hits, h.data(), h.size() * sizeof(uc_hit)); }
    });
}

int uc_engine_hits_set(uc_engine *e, const uint32_t *counts, const uc_hit *hits) {
    return guard([&] { require(e, "engine"); require(counts, "counts"); e->e->set_hits(counts, hits); });
}

int uc_engine_hits_merge(uc_engine *e, int n_parts, const uint32_t *const *counts, const uc_hit *const *hits) {
    return guard([&] {
        require(e, "engine"); require(counts, "counts"); require(hits, "hits");
        std::vector<uint32_t> c;
        std::vector<uc_hit> h;
        merge_hits(e->e->hdb.n, e->e->p.max_seqs, n_parts, counts, hits, c, h);
        e->e->set_hits(c.data(), h.data());
    });
}

int uc_engine_hits_export_dev(const uc_engine *e, uint32_t *d_query, uint32_t *d_target, int32_t *d_score, int32_t *d_diag) {
    return guard([&] {
        require(e, "engine");
        if (e->e->n_hits) { require(d_query, "d_query"); require(d_target, "d_target"); require(d_score, "d_score"); require(d_diag, "d_diag"); }
        e->e->export_hits_dev(d_query, d_target, d_score, d_diag);
    });
}

int uc_engine_hits_import_dev(uc_engine *e, uint64_t n, const uint32_t *d_query, const uint32_t *d_target, const int32_t *d_score,
                              const int32_t *d_diag, uint32_t rank, uint32_t world, uint64_t *n_kept) {
    return guard([&] {
        require(e, "engine");
        if (n) { require(d_query, "d_query"); require(d_target, "d_target"); require(d_score, "d_score"); require(d_diag, "d_diag"); }
        if (world < 1 || rank >= world) fail(UC_ERR_ARGS, "hits_import_dev: rank %u outside world %u", rank, world);
        const uint64_t k = e->e->import_hits_dev(n, d_query, d_target, d_score, d_diag, rank, world);
        if (n_kept) *n_kept = k;
    });
}

int uc_engine_align(uc_engine *e, uint32_t qbegin, uint32_t qend) {
    return guard([&] { require(e, "engine"); e->e->align(qbegin, qend); });
}

int uc_engine_alns_get(const uc_engine *e, uint32_t qbegin, uint32_t qend, uc_aln *out) {
    return guard([&] {
        require(e, "engine");
        const Engine &E = *e->e;
        if (!E.have_db) fail(UC_ERR_ARGS, "alns_get: no database loaded");
        if (qbegin > qend || qend > E.hdb.n) fail(UC_ERR_ARGS, "alns_get: bad query range");
        const uint64_t b = E.hit_off[qbegin], n = E.hit_off[qend] - b;
        if (n) { require(out, "out"); E.get_alns(b, n, out); }
    });
}

int uc_engine_edges_size(const uc_engine *e, uint64_t *n_edges) {
    return guard([&] { require(e, "engine"); require(n_edges, "n_edges"); *n_edges = e->e->edges_on_host ? e->e->edges.size() / 2 : e->e->n_edges_dev; });
}

int uc_engine_edges_get(const uc_engine *e, uint32_t *edges) {
    return guard([&] {
        require(e, "engine");
        const std::vector<uint32_t> &h = e->e->host_edges();
        if (!h.empty()) { require(edges, "edges"); memcpy(edges, h.data(), h.size() * 4); }
    });
}

int uc_engine_stats(const uc_engine *e, uc_stats *out) {
    return guard([&] { require(e, "engine"); require(out, "out"); *out = e->e->stats; });
}

void uc_engine_reset_stats(uc_engine *e) {
    if (!e || !e->e) return;
    uc_stats &s = e->e->stats;
    const uint64_t ns = s.n_seqs, nr = s.n_residues;
    memset(&s, 0, sizeof s);
    s.n_seqs = ns; s.n_residues = nr;
}

int uc_setcover(uint32_t n, const uint32_t *edges, uint64_t n_edges, uint32_t *assign) {
    return guard([&] {
        if (n) require(assign, "assign");
        if (n_edges) require(edges, "edges");
        set_cover(n, edges, n_edges, assign);
    });
}

int uc_engine_setcover(uc_engine *e, const uint32_t *edges, uint64_t n_edges, uint32_t *assign) {
    return guard([&] {
        require(e, "engine");
        const uint32_t n = e->e->hdb.n;
        if (n) require(assign, "assign");
        if (n_edges) require(edges, "edges");
        Timer tc;
        e->e->set_cover_device(n, edges, n_edges, assign);
        e->e->stats.stage_seconds[UC_ST_SETCOVER] += tc.seconds();
    });
}

int uc_write_cluster_db(const char *out_cluster_db, uint32_t n, const uint32_t *assign) {
    return guard([&] {
        require(out_cluster_db, "out_cluster_db");
        std::vector<uint64_t> keys(n);
        for (uint32_t i = 0; i < n; i++) keys[i] = i;
        write_cluster_db(out_cluster_db, keys, assign, n);
    });
}

int uc_engine_ungapped_batch(uc_engine *e, uint64_t n, const uint32_t *q, const uint32_t *t, const int32_t *diag, int32_t *score_out) {
    return guard([&] {
        require(e, "engine");
        if (n) { require(q, "q"); require(t, "t"); require(diag, "diag"); require(score_out, "score_out"); }
        e->e->ungapped_batch(n, q, t, diag, score_out);
    });
}

int uc_engine_sw_batch(uc_engine *e, int mode, uint64_t n, const uint32_t *q, const uint32_t *t, const int32_t *qend_in,
                       const int32_t *tend_in, int32_t *score_out, int32_t *qend_out, int32_t *tend_out) {
    return guard([&] {
        require(e, "engine");
        if (mode < 0 || mode > 2) fail(UC_ERR_ARGS, "sw_batch: mode must be 0, 1 or 2");
        if (!n) return;
        require(q, "q"); require(t, "t"); require(score_out, "score_out");
        if (mode == 2) { require(qend_in, "qend_in"); require(tend_in, "tend_in"); }
        std::vector<PairIn> pairs(n);
        for (uint64_t i = 0; i < n; i++) pairs[i] = {q[i], t[i], mode == 2 ? qend_in[i] : 0, mode == 2 ? tend_in[i] : 0};
        e->e->sw_batch(mode, pairs, score_out, qend_out, tend_out);
    });
}

}  // extern "C"

namespace {

// counters add up over the ranks of a run; times are the slowest rank's (the ranks run side by side)
void merge_stats(uc_stats &d, const uc_stats &s) {
    d.n_index_entries += s.n_index_entries; d.n_sim_kmers += s.n_sim_kmers; d.n_kmer_hits += s.n_kmer_hits;
    d.n_candidates += s.n_candidates; d.n_prefilter_hits += s.n_prefilter_hits; d.n_gapped_alignments += s.n_gapped_alignments;
    d.n_start_alignments += s.n_start_alignments; d.n_pk_reruns += s.n_pk_reruns;
    d.cells_fwd += s.cells_fwd; d.cells_rev += s.cells_rev; d.cells_start += s.cells_start; d.cells_tb += s.cells_tb;
    d.sw_kernel_launches += s.sw_kernel_launches; d.sw_algorithmic_bytes += s.sw_algorithmic_bytes;
    d.n_filtered_hits += s.n_filtered_hits; d.n_sw_runs += s.n_sw_runs; d.cells_run += s.cells_run; d.exchange_bytes += s.exchange_bytes;
    for (int k = 0; k < UC_NSTAGE; k++) {
        if (k != UC_ST_SETCOVER && k != UC_ST_OUTPUT && k != UC_ST_LOAD) d.algorithmic_bytes[k] += s.algorithmic_bytes[k];
        d.stage_seconds[k] = std::max(d.stage_seconds[k], s.stage_seconds[k]);
    }
    d.sw_kernel_ms = std::max(d.sw_kernel_ms, s.sw_kernel_ms);
    d.prefilter_kernel_ms = std::max(d.prefilter_kernel_ms, s.prefilter_kernel_ms);
    d.exchange_seconds = std::max(d.exchange_seconds, s.exchange_seconds);
    if (s.phase_seconds[3] > d.phase_seconds[3]) memcpy(d.exchange2_seconds, s.exchange2_seconds, sizeof d.exchange2_seconds);   // the split of the slowest rank's phase 3
    for (int k = 0; k < UC_NPHASE; k++) d.phase_seconds[k] = std::max(d.phase_seconds[k], s.phase_seconds[k]);
}

// the sub-database of the current representatives (createsubdb)
HostDb sub_db(const HostDb &full, const std::vector<uint32_t> &cur) {
    HostDb sub;
    sub.n = (uint32_t)cur.size();
    sub.off.resize((size_t)sub.n + 1);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < sub.n; i++) { sub.off[i] = tot; tot += full.len(cur[i]); }
    sub.off[sub.n] = tot;
    sub.s3.resize(tot); sub.sa.resize(tot); sub.keys.resize(sub.n);
    for (uint32_t i = 0; i < sub.n; i++) {
        memcpy(sub.s3.data() + sub.off[i], full.s3.data() + full.off[cur[i]], full.len(cur[i]));
        memcpy(sub.sa.data() + sub.off[i], full.sa.data() + full.off[cur[i]], full.len(cur[i]));
        sub.keys[i] = full.keys[cur[i]];
    }
    return sub;
}

}  // namespace

extern "C" {

// ---- the three calls of /root/reference/src/modules/cluster.rs:45-76 ----------------------------

int uc_cluster(const char *db, const char *out_cluster_db, const char *tmp, const uc_opts *o, uc_stats *stats_out) {
    return guard([&] {
        require(db, "db"); require(out_cluster_db, "out_cluster_db");
        const bool stamp = getenv("UC_TIMING") != nullptr;
        auto mark = [&](const char *what) { if (stamp) fprintf(stderr, "unicore-cluster[timing]: t+%7.1f ms  %s\n", 1e3 * g_library_loaded.seconds(), what); };
        mark("uc_cluster entered");
        Params p = params_from(o);
        mark("options + matrices");
        if (tmp && *tmp) mkdir_p(tmp);   // callee owns <tmp>; nothing is spilled there yet
        if (p.mat3di_synthetic)
            logf(1, "Warning: clustering with the SYNTHETIC 3Di matrix %s (UC_ALLOW_SYNTHETIC=1): results are not Foldseek's\n", p.mat3di_path.c_str());

        // The HIP runtime and the device context take 0.15-0.3 s to come up in a fresh process (tools/cold_stamps.sh): they are initialised on
        // a helper thread WHILE this thread reads and encodes the database (a one-shot `foldseek cluster` process pays for both every time).
        const int want_dev = o && o->device >= 0 ? o->device : 0;
        std::future<int> warm = std::async(std::launch::async, [want_dev]() -> int {
            int nd = 0;
            if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return 0;
            if (hipSetDevice(want_dev < nd ? want_dev : 0) == hipSuccess) (void)hipFree(nullptr);      // forces the primary context into existence
            return nd;
        });
        Timer tl;
        logf(3, "unicore-cluster: reading %s\n", db);
        HostDb full;
        read_seq_db(db, full, false);
        const double t_load = tl.seconds();
        mark("database read + encoded");
        // ---- devices (one engine + one host thread per GPU, hit lists all-gathered with RCCL)
        const int ndev = warm.get();
        if (ndev <= 0) fail(UC_ERR_DEVICE, "no HIP device available; this engine has no CPU fallback");
        mark("HIP runtime + device context ready (initialised beside the database read)");
        int W = p.num_gpus == 0 ? ndev : p.num_gpus;
        std::vector<int> devices;
        bool virtual_gpus = false;
        if (W == 1) {
            int d = o ? o->device : -1;
            if (d < 0) UC_HIP(hipGetDevice(&d));
            devices.push_back(d);
        } else {
            if (W > ndev) {
                // several ranks per physical GPU: only to exercise the N > 1 path on a single-GPU box (tests); RCCL refuses
                // two ranks on one device, so these ranks exchange with plain device copies
                const char *v = getenv("UC_VIRTUAL_GPUS");
                if (!v || strcmp(v, "1") != 0) fail(UC_ERR_DEVICE, "%d GPUs requested but only %d visible", W, ndev);
                virtual_gpus = true;
            }
            for (int r = 0; r < W; r++) devices.push_back(r % ndev);
        }
        int gQ = 1, gT = 1;
        grid_shape(W, p.target_shards, &gQ, &gT);


        const uint32_t n = full.n;
        if (n == 0) {   // an empty, well-formed database: an empty cluster DB, whatever the workflow
            write_cluster_db(out_cluster_db, full.keys, nullptr, 0);
            logf(3, "unicore-cluster: empty database -> %s\n", out_cluster_db);
            if (stats_out) { *stats_out = uc_stats(); stats_out->n_gpus = (uint32_t)W; stats_out->target_shards = (uint32_t)gT; stats_out->stage_seconds[UC_ST_LOAD] = t_load; }
            return;
        }
        const int pre = p.linclust ? 1 : 0;     // E8a: one linear-time pre-clustering round in front of the cascade rounds
        // the kernels' code objects are uploaded on a helper thread beside engine construction, database upload and prefilter (uc_sw.hip:preload_modules);
        // (The future's destructor joins the thread on every way out of this function.)
        std::future<void> preload;
        if (W == 1) {
            const int d0 = devices[0];
            const bool lc = pre != 0, st = stamp;
            preload = std::async(std::launch::async, [d0, lc, st]() {
                preload_modules(d0, lc);
                if (st) fprintf(stderr, "unicore-cluster[timing]: t+%7.1f ms  (helper thread) code objects of the prefilter / alignment / SW modules loaded\n", 1e3 * g_library_loaded.seconds());
            });
        }
        logf(3, "unicore-cluster: %u sequences, %llu residues, %d GPU(s)%s [%d query group(s) x %d target shard(s)], %s%d clustering step(s)\n", n,
             (unsigned long long)full.residues(), W, virtual_gpus ? " (virtual: several ranks per device)" : "", gQ, gT,
             pre ? "linear-time pre-step + " : "", p.cluster_steps);

        // ---- state shared by the rank threads (host memory; rank 0 owns the workflow, the others follow its rounds)
        std::vector<uint32_t> assign(n), cur(n), posmap(n);
        for (uint32_t i = 0; i < n; i++) { assign[i] = i; cur[i] = i; }
        HostDb round_db;                       // sub-database of the round (rounds > 0)
        std::vector<uint32_t> pre_pairs;       // candidate pairs of the pre-step
        int round_kmer_thr = p.kmer_thr;
        LocalGroup grp(W);
        if (virtual_gpus && getenv("UC_VIRTUAL_SERIAL")) grp.serialize = true;   // emulation: one rank's compute phase at a time on the shared GPU
        std::vector<std::unique_ptr<Comm>> comms;
        for (int r = 0; r < W; r++) {
            comms.emplace_back(new Comm);
            comms.back()->rank = r; comms.back()->world = W; comms.back()->grp = W > 1 ? &grp : nullptr;
        }
        std::vector<uc_stats> rank_stats((size_t)W);
        uint64_t n_clusters = 0;
        uint32_t nccl_ranks = 0;

        auto rank_main = [&](int r) {
            Engine E(p, devices[(size_t)r]);
            if (r == 0) mark("engine constructed (streams, events, matrices on the device)");
            Comm &C = *comms[(size_t)r];
            // test hook: UC_FAIL_RANK="<rank>:<stage>" makes that rank throw (stage 0 = right after its engine exists, 1 = between
            // its prefilter and the exchange) — the failure paths below must turn that into an error code, never into a hang
            int fail_rank = -1, fail_stage = -1;
            if (const char *fr = getenv("UC_FAIL_RANK")) sscanf(fr, "%d:%d", &fail_rank, &fail_stage);
            if (fail_rank == r && fail_stage == 0) fail(UC_ERR_DEVICE, "injected failure of rank %d (UC_FAIL_RANK)", r);
            if (W > 1) {
                // every engine exists (a constructor that threw has failed the group: this barrier then throws on the others)
                // BEFORE any rank enters RCCL; rank 0 then creates all communicators in one call — a collective init per rank
                // would leave the survivors of a failed peer blocked inside ncclCommInitRank for good
                grp.barrier();
                if (r == 0 && !virtual_gpus) {
                    std::vector<Comm *> cs;
                    for (auto &c : comms) cs.push_back(c.get());
                    comm_init_all(cs, devices);
                    int cnt = 0, rk = 0, dv = 0;
                    comm_info(C, &cnt, &rk, &dv);
                    nccl_ranks = (uint32_t)cnt;
                    logf(3, "unicore-cluster: RCCL communicator over %d ranks (ncclCommCount)\n", cnt);
                }
                grp.barrier();
            }
            const bool timing = getenv("UC_TIMING") != nullptr;
            Timer tph;
            auto phase = [&](const char *what, int rr) {
                if (timing && r == 0) fprintf(stderr, "unicore-cluster[timing]: t+%7.1f ms  round %d %-14s %.1f ms\n", 1e3 * g_library_loaded.seconds(), rr, what, 1e3 * tph.seconds());
                tph = Timer();
            };
            for (int rr = 0; rr < p.cluster_steps + pre; rr++) {
                const int rd = rr - pre;             // cascade round index (-1 = the pre-step)
                phase("(between)", rr);
                if (W == 1 && rr == 1) full = std::move(E.hdb);      // a single rank borrows the databases instead of copying them
                const bool dev_sub = W == 1 && rr > 0;                // ... and lays the representatives' sub-database out on the device
                if (r == 0) {
                    if (rr > 0 && !dev_sub) round_db = sub_db(full, cur);
                    round_kmer_thr = p.kmer_thr;
                    if (rd >= 0 && p.cluster_steps > 1 && !p.kmer_thr_explicit)   // sensitivity rises linearly from 1 to the target (spec UC-1 E8)
                        round_kmer_thr = kmer_thr_for(p, 1.0 + (p.sensitivity - 1.0) * rd / (p.cluster_steps - 1));
                }
                C.barrier(E);
                phase("sub_db", rr);
                if (!dev_sub) {
                    if (W == 1) E.hdb = std::move(rr == 0 ? full : round_db);
                    else E.hdb = rr == 0 ? full : round_db;
                }
                phase("hdb copy", rr);
                E.p.kmer_thr = round_kmer_thr;
                if (dev_sub) E.upload_sub_db(cur, full.off);
                else E.upload_db(/*keep_raw=*/W == 1 && p.cluster_steps + pre > 1);
                phase("upload", rr);
                const uint32_t m = E.hdb.n;
                if (rd < 0 && W == 1) {   // E8a on one rank: the candidate pairs become the hit lists without leaving the device
                    const uint64_t np = E.linclust_hits();
                    phase("linclust_pairs", rr);
                    E.stats.n_prefilter_hits += np;
                    logf(3, "unicore-cluster: pre-step: %u sequences, %llu candidate pairs (%d k-mers per sequence)\n", m, (unsigned long long)np, p.kmer_per_seq);
                } else if (rd < 0) {   // E8a: candidate pairs (centre, member) from shared minimum-hash k-mers, the centre is the query
                    if (r == 0) pre_pairs = E.linclust_pairs();
                    phase("linclust_pairs", rr);
                    C.barrier(E);
                    // a rank aligns the pairs of the centres it owns (centre mod world): whole queries stay together
                    std::vector<uint32_t> cnt(m, 0);
                    std::vector<uc_hit> hl;
                    hl.reserve(pre_pairs.size() / 2 / (size_t)W + 16);
                    for (size_t k = 0; k < pre_pairs.size() / 2; k++) {
                        const uint32_t c = pre_pairs[2 * k];
                        if ((int)(c % (uint32_t)W) != r) continue;
                        cnt[c]++;
                        uc_hit h; h.target = pre_pairs[2 * k + 1]; h.score = 0; h.diag = 0;
                        hl.push_back(h);
                    }
                    E.set_hits(cnt.data(), hl.data(), /*check_max_seqs=*/false);
                    E.stats.n_prefilter_hits += hl.size();
                    if (r == 0)
                        logf(3, "unicore-cluster: pre-step: %u sequences, %zu candidate pairs (%d k-mers per sequence)\n", m, pre_pairs.size() / 2, p.kmer_per_seq);
                } else {
                    {
                        Turn turn(C, &E);
                        Timer tp;
                        prefilter_cell(E, W, p.target_shards, r);
                        E.stats.phase_seconds[0] += tp.seconds();
                    }
                    if (fail_rank == r && fail_stage == 1) fail(UC_ERR_DEVICE, "injected failure of rank %d (UC_FAIL_RANK)", r);
                    if (W > 1) exchange_hits(E, C);
                    if (r == 0)
                        logf(3, "unicore-cluster: step %d: %u sequences, prefilter kept %llu pairs%s (k-score %d, max-seqs %d)\n", rd + 1, m,
                             (unsigned long long)E.n_hits, W > 1 ? " on rank 0" : "", E.p.kmer_thr, p.max_seqs);
                }
                phase("hits", rr);
                {
                    Turn turn(C, &E);
                    Timer ta;
                    E.align(0, m);
                    E.stats.phase_seconds[5] += ta.seconds();
                }
                phase("align", rr);
                if (r == 0) call_round_hook(E, rd, cur);
                Timer te;
                uint64_t n_acc = 0;                  // accepted pairs of the round (all ranks)
                const uint32_t *dev_all = nullptr;   // rank 0: every rank's edges in one device buffer (gathered device to device)
                if (W > 1) { n_acc = C.gather_edges_dev(E, &dev_all); E.stats.exchange_seconds += te.seconds(); }
                else n_acc = E.edges_on_host ? E.edges.size() / 2 : E.n_edges_dev;
                if (r == 0) {
                    Timer tc;
                    std::vector<uint32_t> sa(m);
                    if (W > 1 && n_acc < (1ull << 31)) E.set_cover_graph(m, nullptr, dev_all, n_acc, sa.data());
                    else if (W > 1) {   // beyond the 32-bit positions of the device graph build: the all-host cover
                        std::vector<uint32_t> all(2 * n_acc);
                        UC_HIP(hipMemcpy(all.data(), dev_all, 2 * n_acc * 4, hipMemcpyDeviceToHost));
                        set_cover(m, all.data(), n_acc, sa.data());
                    }
                    else E.set_cover_own_edges(m, sa.data());      // one rank: graph straight from the device-resident edge list
                    // mergeclusters: the representative of a sequence is the representative of its representative
                    for (uint32_t i = 0; i < cur.size(); i++) posmap[cur[i]] = i;
                    for (uint32_t x = 0; x < n; x++) assign[x] = cur[sa[posmap[assign[x]]]];
                    std::vector<uint32_t> next;
                    for (uint32_t i = 0; i < cur.size(); i++) if (sa[i] == i) next.push_back(cur[i]);
                    E.stats.algorithmic_bytes[UC_ST_SETCOVER] += 8ull * n_acc + 4ull * m;
                    E.stats.stage_seconds[UC_ST_SETCOVER] += tc.seconds();
                    if (W > 1) E.stats.phase_seconds[7] += tc.seconds();
                    logf(3, "unicore-cluster: step %d: %llu accepted pairs, %zu representatives\n", rd + 1, (unsigned long long)n_acc, next.size());
                    cur.swap(next);
                    E.stats.n_edges = n_acc;
                }
                phase("cover+merge", rr);
            }
            C.barrier(E);
            if (W == 1 && p.cluster_steps + pre == 1) full = std::move(E.hdb);
            if (r == 0) n_clusters = cur.size();
            rank_stats[(size_t)r] = E.stats;
        };

        if (W == 1) rank_main(0);
        else {
            std::vector<std::string> err((size_t)W);
            std::vector<int> code((size_t)W, 0);
            std::vector<std::thread> th;
            // a rank on its way out fails the group (peers leave their next barrier with an error) and aborts every RCCL
            // communicator (peers blocked INSIDE a collective return from it)
            auto bail = [&] { grp.fail_all(); std::vector<Comm *> cs; for (auto &c : comms) cs.push_back(c.get()); abort_all(cs); };
            for (int r = 0; r < W; r++)
                th.emplace_back([&, r] {
                    try { rank_main(r); }
                    catch (const Error &e) { code[(size_t)r] = e.code; err[(size_t)r] = e.what(); bail(); }
                    catch (const std::exception &e) { code[(size_t)r] = UC_ERR_GENERIC; err[(size_t)r] = e.what(); bail(); }
                });
            for (auto &t : th) t.join();
            // report the rank that failed first-hand, not the ones that were released from a barrier because of it
            int bad = -1;
            for (int r = 0; r < W; r++)
                if (code[(size_t)r] && (bad < 0 || err[(size_t)bad].find("another GPU rank") != std::string::npos)) bad = r;
            if (bad >= 0) fail(code[(size_t)bad], "GPU rank %d: %s", bad, err[(size_t)bad].c_str());
        }

        uc_stats st = rank_stats[0];
        for (int r = 1; r < W; r++) merge_stats(st, rank_stats[(size_t)r]);
        st.stage_seconds[UC_ST_LOAD] += t_load;
        st.n_clusters = n_clusters;
        st.n_seqs = n;
        st.n_residues = full.residues();
        st.n_gpus = (uint32_t)W;
        st.target_shards = (uint32_t)gT;
        st.nccl_ranks = nccl_ranks;
        Timer to;
        write_cluster_db(out_cluster_db, full.keys, assign.data(), n);
        st.stage_seconds[UC_ST_OUTPUT] += to.seconds();
        mark("cluster DB written");
        logf(3, "unicore-cluster: %llu clusters -> %s\n", (unsigned long long)n_clusters, out_cluster_db);
        if (stats_out) *stats_out = st;
    });
}

int uc_comm_unique_id(uint8_t id[UC_COMM_ID_BYTES]) {
    return guard([&] { require(id, "id"); comm_unique_id(id); });
}

int uc_comm_create(const uint8_t id[UC_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, uc_comm **out) {
    return guard([&] {
        require(id, "id"); require(out, "out");
        *out = nullptr;
        if (world < 1 || rank < 0 || rank >= world) fail(UC_ERR_ARGS, "uc_comm_create: rank %d outside world %d", rank, world);
        auto h = std::make_unique<uc_comm>();
        if (device < 0) UC_HIP(hipGetDevice(&device));
        comm_init_rank(h->c, id, rank, world, device);
        *out = h.release();
    });
}

void uc_comm_destroy(uc_comm *c) { delete c; }

int uc_comm_info(const uc_comm *c, int32_t *nccl_ranks, int32_t *nccl_rank, int32_t *device) {
    return guard([&] {
        require(c, "comm");
        int n = 0, r = 0, d = 0;
        comm_info(c->c, &n, &r, &d);
        if (nccl_ranks) *nccl_ranks = n;
        if (nccl_rank) *nccl_rank = r;
        if (device) *device = d;
    });
}

int uc_engine_cluster_step(uc_engine *e, uc_comm *comm, int32_t target_shards, uint32_t *assign, uint64_t *n_alignments) {
    return guard([&] {
        require(e, "engine");
        Comm solo;
        Comm &C = comm ? comm->c : solo;
        const uint64_t k = cluster_step(*e->e, C, target_shards, assign);
        if (n_alignments) *n_alignments = k;
    });
}

int uc_search(const char *query_db, const char *target_db, const char *out_aln_db, const char *tmp, const uc_opts *o, uc_stats *stats_out) {
    return guard([&] {
        require(query_db, "query_db"); require(target_db, "target_db"); require(out_aln_db, "out_aln_db");
        Params p = params_from(o, true);
        p.want_tb = 1;
        if (tmp && *tmp) mkdir_p(tmp);
        Engine E(p, o ? o->device : -1);
        Timer tl;
        HostDb T, Q;
        read_seq_db(target_db, T, false);
        read_seq_db(query_db, Q, false);
        {   // 3Di tracks predicted under different readings of the ProstT5 head (uc_createdb records its reading in <db>_ss.source) are not comparable at the sequence ends
            auto source_of = [](const std::string &db) { std::ifstream f(db + "_ss.source"); std::string l; if (f) std::getline(f, l); return l; };
            const std::string st = source_of(target_db), sq = source_of(query_db);
            if (!st.empty() && !sq.empty() && st != sq)
                logf(2, "unicore-search: WARNING: the 3Di tracks of the two databases were predicted under different ProstT5 head conventions (%s: '%s'; %s: '%s')\n",
                     target_db, st.c_str(), query_db, sq.c_str());
        }
        const uint32_t nt = T.n, nq = Q.n, n = nt + nq;
        if ((uint64_t)nt + nq >= (1u << 24)) fail(UC_ERR_ARGS, "this build supports < 2^24 sequences (query + target)");
        // one loaded set: targets first, then queries; only [0, nt) is indexed, only [nt, n) are queries
        HostDb &C = E.hdb;
        C.n = n;
        C.keys = T.keys; C.keys.insert(C.keys.end(), Q.keys.begin(), Q.keys.end());
        C.off.resize((size_t)n + 1);
        const uint64_t rt = T.residues();
        for (uint32_t i = 0; i <= nt; i++) C.off[i] = T.off[i];
        for (uint32_t i = 0; i <= nq; i++) C.off[nt + i] = rt + Q.off[i];
        C.s3 = T.s3; C.s3.insert(C.s3.end(), Q.s3.begin(), Q.s3.end());
        C.sa = T.sa; C.sa.insert(C.sa.end(), Q.sa.begin(), Q.sa.end());
        E.stats.stage_seconds[UC_ST_LOAD] += tl.seconds();
        E.evalue_residues = rt;
        E.upload_db();
        logf(3, "unicore-search: %u queries vs %u targets (%llu residues) on device %d\n", nq, nt, (unsigned long long)rt, E.device);
        E.prefilter(0, nt, nt, n);
        logf(3, "unicore-search: prefilter kept %llu pairs (k-score %d, max-seqs %d)\n", (unsigned long long)E.n_hits, p.kmer_thr, p.max_seqs);
        E.align(nt, n);
        Timer to;
        std::vector<uc_hit> hits(std::max<uint64_t>(E.n_hits, 1));
        std::vector<uc_aln> alns(std::max<uint64_t>(E.n_hits, 1));
        if (E.n_hits) { E.get_hits(hits.data()); E.get_alns(0, E.n_hits, alns.data()); }
        std::vector<std::vector<AlnRow>> rows(nq);
        uint64_t n_acc = 0;
        for (uint32_t q = 0; q < nq; q++) {
            const uint32_t qi = nt + q;
            const int lq = (int)C.len(qi);
            std::vector<AlnRow> &rv = rows[q];
            for (uint64_t k = E.hit_off[qi]; k < E.hit_off[qi + 1]; k++) {
                const uc_aln &a = alns[k];
                if (!a.accepted) continue;
                const uint32_t t = hits[k].target;
                AlnRow r;
                r.tkey = T.keys[t];
                r.corrected = a.corrected;
                r.bits = (int32_t)((p.lambda * (double)a.corrected - std::log(p.Kconst)) / std::log(2.0));
                r.evalue = p.Kconst * (double)lq * (double)rt * std::exp(-p.lambda * (double)a.corrected);
                r.fident = a.aln_len > 0 ? (double)a.idents / (double)a.aln_len : 0.0;
                r.qstart = a.qstart; r.qend = a.qend; r.qlen = lq; r.tstart = a.tstart; r.tend = a.tend; r.tlen = (int32_t)T.len(t);
                r.aln_len = a.aln_len; r.idents = a.idents; r.gap_opens = a.gap_opens;
                rv.push_back(r);
            }
            std::sort(rv.begin(), rv.end(), [](const AlnRow &x, const AlnRow &y) { return x.corrected != y.corrected ? x.corrected > y.corrected : x.tkey < y.tkey; });
            n_acc += rv.size();
        }
        write_aln_db(out_aln_db, Q.keys, rows);
        E.stats.stage_seconds[UC_ST_OUTPUT] += to.seconds();
        logf(3, "unicore-search: %llu alignments, %llu accepted -> %s\n", (unsigned long long)E.stats.n_gapped_alignments, (unsigned long long)n_acc, out_aln_db);
        E.stats.n_seqs = n;
        if (stats_out) *stats_out = E.stats;
    });
}

int uc_convertalis(const char *query_db, const char *target_db, const char *aln_db, const char *out_m8, const uc_opts *o) {
    return guard([&] {
        require(query_db, "query_db"); require(target_db, "target_db"); require(aln_db, "aln_db"); require(out_m8, "out_m8");
        if (o) g_verbosity = o->verbosity;
        convert_alis(query_db, target_db, aln_db, out_m8);
    });
}

int uc_createtsv(const char *db, const char *cluster_db, const char *out_tsv, const uc_opts *o) {
    return guard([&] {
        require(db, "db"); require(cluster_db, "cluster_db"); require(out_tsv, "out_tsv");
        if (o) g_verbosity = o->verbosity;
        create_tsv(db, cluster_db, out_tsv);
    });
}

int uc_createdb(const char *const *fasta_paths, int n_fasta, const char *out_db, const char *model, const uc_opts *o, uc_t5_stats *stats_out) {
    return guard([&] {
        require(fasta_paths, "fasta_paths"); require(out_db, "out_db"); require(model, "model");
        if (n_fasta < 1) fail(UC_ERR_ARGS, "createdb: no input files");
        std::vector<std::string> fp;
        for (int i = 0; i < n_fasta; i++) { require(fasta_paths[i], "fasta path"); fp.push_back(fasta_paths[i]); }
        // devices: the convention of uc_cluster (uc_opts.num_gpus: 0 = all visible, 1 = `device`, N = devices 0..N-1; more than are visible only with
        // UC_VIRTUAL_GPUS=1: several replicas per device, the single-GPU box's test of this path).  The encoder shards by sequence, "replicas only".
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) fail(UC_ERR_DEVICE, "no HIP device available; the ProstT5 encoder has no CPU fallback");
        const int want = o ? o->num_gpus : 1;
        if (want < 0 || want > 64) fail(UC_ERR_ARGS, "createdb: num_gpus must be in [0,64] (0 = all visible)");
        const int W = want == 0 ? ndev : want;
        std::vector<int> devices;
        if (W == 1) {
            int d = o ? o->device : -1;
            if (d < 0) UC_HIP(hipGetDevice(&d));
            devices.push_back(d);
        } else {
            if (W > ndev) {
                const char *v = getenv("UC_VIRTUAL_GPUS");
                if (!v || strcmp(v, "1") != 0) fail(UC_ERR_DEVICE, "%d GPUs requested but only %d visible", W, ndev);
            }
            for (int r = 0; r < W; r++) devices.push_back(r % ndev);
        }
        T5Stats st;
        std::vector<T5Stats> per;
        t5_createdb(fp, out_db, model, devices, o ? o->verbosity : 3, &st, &per);
        if (stats_out) {
            stats_out->n_seqs = st.n_seqs; stats_out->n_tokens = st.n_tokens; stats_out->flops = st.flops; stats_out->gpu_ms = st.total_ms;
            stats_out->n_replicas = (uint32_t)per.size(); stats_out->reserved0 = 0;
            stats_out->gpu_ms_sum = 0;
            uint64_t lo = UINT64_MAX, hi = 0;
            for (const T5Stats &x : per) { stats_out->gpu_ms_sum += x.total_ms; lo = std::min<uint64_t>(lo, x.n_tokens); hi = std::max<uint64_t>(hi, x.n_tokens); }
            stats_out->tokens_min_replica = per.empty() ? 0 : lo; stats_out->tokens_max_replica = hi;
        }
    });
}

int uc_t5_load(const char *model, int32_t device, uc_t5 **out) {
    return guard([&] {
        require(model, "model"); require(out, "out");
        *out = nullptr;
        std::string path = model;
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) path += "/prostt5-f16.gguf";
        auto h = std::make_unique<uc_t5>();
        h->m.load(path, device);
        *out = h.release();
    });
}

void uc_t5_free(uc_t5 *m) { delete m; }

int uc_t5_encode(uc_t5 *m, uint32_t n, const uint64_t *off, const char *aa, uint8_t *codes, float *logits) {
    return guard([&] {
        require(m, "model");
        if (!n) return;
        require(off, "off"); require(aa, "aa"); require(codes, "codes");
        std::vector<std::string> seqs(n);
        for (uint32_t i = 0; i < n; i++) {
            if (off[i + 1] < off[i]) fail(UC_ERR_ARGS, "t5_encode: offsets must be non-decreasing");
            seqs[i].assign(aa + off[i], aa + off[i + 1]);
        }
        std::vector<std::vector<uint8_t>> out;
        std::vector<std::vector<float>> lg;
        m->m.encode(seqs, out, logits ? &lg : nullptr);
        for (uint32_t i = 0; i < n; i++) {
            if (!out[i].empty()) memcpy(codes + off[i], out[i].data(), out[i].size());
            if (logits && !lg[i].empty()) memcpy(logits + off[i] * (size_t)m->m.cfg.n_out, lg[i].data(), lg[i].size() * 4);
        }
    });
}

int uc_t5_get_stats(const uc_t5 *m, uc_t5_stats *out) {
    return guard([&] {
        require(m, "model"); require(out, "out");
        out->n_seqs = m->m.stats.n_seqs; out->n_tokens = m->m.stats.n_tokens; out->flops = m->m.stats.flops; out->gpu_ms = m->m.stats.total_ms;
        out->n_replicas = 1; out->reserved0 = 0; out->gpu_ms_sum = m->m.stats.total_ms; out->tokens_min_replica = out->tokens_max_replica = m->m.stats.n_tokens;
    });
}

int uc_rmdb(const char *db_prefix) {
    return guard([&] { require(db_prefix, "db_prefix"); remove_db(db_prefix); });
}

}  // extern "C"
