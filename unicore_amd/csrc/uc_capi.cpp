// uc_capi.cpp — extern "C" boundary (include/unicore_cluster.h).  No exception crosses it: every entry
// point converts failures into the status classes of the header and records uc_last_error().
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include "uc_engine.h"

using namespace uc;

struct uc_engine {
    std::unique_ptr<Engine> e;
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

void require(const void *p, const char *what) {
    if (!p) fail(UC_ERR_ARGS, "%s must not be NULL", what);
}

}  // namespace

extern "C" {

const char *uc_last_error(void) { return last_error_cstr(); }
const char *uc_version(void) { return "unicore-cluster-mi355x 0.1.0 (spec UC-1, gfx950)"; }

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
        if (qbegin > qend || qend > E.hdb.n) fail(UC_ERR_ARGS, "alns_get: bad query range");
        const uint64_t b = E.hit_off[qbegin], n = E.hit_off[qend] - b;
        if (n) { require(out, "out"); E.get_alns(b, n, out); }
    });
}

int uc_engine_edges_size(const uc_engine *e, uint64_t *n_edges) {
    return guard([&] { require(e, "engine"); require(n_edges, "n_edges"); *n_edges = e->e->edges.size() / 2; });
}

int uc_engine_edges_get(const uc_engine *e, uint32_t *edges) {
    return guard([&] {
        require(e, "engine");
        if (!e->e->edges.empty()) { require(edges, "edges"); memcpy(edges, e->e->edges.data(), e->e->edges.size() * 4); }
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

// ---- the three calls of /root/reference/src/modules/cluster.rs:45-76 ----------------------------

int uc_cluster(const char *db, const char *out_cluster_db, const char *tmp, const uc_opts *o, uc_stats *stats_out) {
    return guard([&] {
        require(db, "db"); require(out_cluster_db, "out_cluster_db");
        Params p = params_from(o);
        if (tmp && *tmp) mkdir_p(tmp);   // callee owns <tmp>; nothing is spilled there yet
        Engine E(p, o ? o->device : -1);
        Timer tl;
        logf(3, "unicore-cluster: reading %s\n", db);
        read_seq_db(db, E.hdb, false);
        E.stats.stage_seconds[UC_ST_LOAD] += tl.seconds();
        const HostDb full = E.hdb;            // round 0 runs on the whole DB; later rounds on representatives (createsubdb)
        const uint32_t n = full.n;
        logf(3, "unicore-cluster: %u sequences, %llu residues, device %d, %d clustering step(s)\n", n, (unsigned long long)full.residues(),
             E.device, p.cluster_steps);
        std::vector<uint32_t> assign(n), cur(n), posmap(n);
        for (uint32_t i = 0; i < n; i++) { assign[i] = i; cur[i] = i; }
        const int pre = p.linclust ? 1 : 0;     // E8a: one linear-time pre-clustering round in front of the cascade rounds
        for (int rr = 0; rr < p.cluster_steps + pre; rr++) {
            const int r = rr - pre;             // cascade round index (-1 = the pre-step)
            if (rr > 0) {   // sub-database of the current representatives
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
                E.hdb = std::move(sub);
            }
            if (r >= 0 && p.cluster_steps > 1 && !p.kmer_thr_explicit)   // sensitivity rises linearly from 1 to the target (spec UC-1 E8)
                E.p.kmer_thr = kmer_thr_for(p, 1.0 + (p.sensitivity - 1.0) * r / (p.cluster_steps - 1));
            E.upload_db();
            if (r < 0) {   // E8a: candidate pairs (centre, member) from shared minimum-hash k-mers, the centre is the query
                const std::vector<uint32_t> pr = linclust_pairs(E.hdb, p, p.threads);
                std::vector<uint32_t> cnt(E.hdb.n, 0);
                std::vector<uc_hit> hl(pr.size() / 2);
                for (size_t k = 0; k < pr.size() / 2; k++) { cnt[pr[2 * k]]++; hl[k].target = pr[2 * k + 1]; hl[k].score = 0; hl[k].diag = 0; }
                E.set_hits(cnt.data(), hl.data(), /*check_max_seqs=*/false);
                E.stats.n_prefilter_hits += pr.size() / 2;
                logf(3, "unicore-cluster: pre-step: %u sequences, %zu candidate pairs (%d k-mers per sequence)\n", E.hdb.n, pr.size() / 2, p.kmer_per_seq);
            } else {
                E.prefilter(0, E.hdb.n);
                logf(3, "unicore-cluster: step %d: %u sequences, prefilter kept %llu pairs (k-score %d, max-seqs %d)\n", r + 1, E.hdb.n,
                     (unsigned long long)E.n_hits, E.p.kmer_thr, p.max_seqs);
            }
            E.align(0, E.hdb.n);
            Timer tc;
            std::vector<uint32_t> sa(E.hdb.n);
            E.set_cover_device(E.hdb.n, E.edges.data(), E.edges.size() / 2, sa.data());
            // mergeclusters: the representative of a sequence is the representative of its representative
            for (uint32_t i = 0; i < cur.size(); i++) posmap[cur[i]] = i;
            for (uint32_t x = 0; x < n; x++) assign[x] = cur[sa[posmap[assign[x]]]];
            std::vector<uint32_t> next;
            for (uint32_t i = 0; i < cur.size(); i++) if (sa[i] == i) next.push_back(cur[i]);
            E.stats.algorithmic_bytes[UC_ST_SETCOVER] += 8ull * (E.edges.size() / 2) + 4ull * E.hdb.n;
            E.stats.stage_seconds[UC_ST_SETCOVER] += tc.seconds();
            logf(3, "unicore-cluster: step %d: %llu accepted pairs, %zu representatives\n", r + 1, (unsigned long long)(E.edges.size() / 2), next.size());
            cur.swap(next);
        }
        E.stats.n_clusters = cur.size();
        E.stats.n_seqs = n;
        E.stats.n_residues = full.residues();
        const uint64_t ncl = cur.size();
        Timer to;
        write_cluster_db(out_cluster_db, full.keys, assign.data(), n);
        E.stats.stage_seconds[UC_ST_OUTPUT] += to.seconds();
        logf(3, "unicore-cluster: %llu clusters -> %s\n", (unsigned long long)ncl, out_cluster_db);
        if (stats_out) *stats_out = E.stats;
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

int uc_rmdb(const char *db_prefix) {
    return guard([&] { require(db_prefix, "db_prefix"); remove_db(db_prefix); });
}

}  // extern "C"
