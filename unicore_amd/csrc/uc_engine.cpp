// uc_engine.cpp — host orchestration of the per-GPU engine: DB upload, batching of the gapped stage
// (E5/E6), kernel-level entry points.  Prefilter orchestration (E1-E4) lives in uc_prefilter.hip.
#include "uc_engine.h"

#include <algorithm>
#include <cstring>
#include <numeric>

namespace uc {

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
    UC_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    UC_HIP(hipEventCreate(&ev0));
    UC_HIP(hipEventCreate(&ev1));
    d_S3.reserve(A * A);
    d_SA.reserve(A * A);
    UC_HIP(hipMemcpy(d_S3.p, p.S3, A * A, hipMemcpyHostToDevice));
    UC_HIP(hipMemcpy(d_SA.p, p.SA, A * A, hipMemcpyHostToDevice));
    memset(&stats, 0, sizeof stats);
}

Engine::~Engine() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
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

void Engine::upload_db() {
    Timer tm;
    UC_HIP(hipSetDevice(device));
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
    std::vector<uint8_t> p3(tot + 64, 20), pa(tot + 64, 20);
    for (uint32_t i = 0; i < n; i++) {
        memcpy(p3.data() + h_poff[i], hdb.s3.data() + hdb.off[i], h_len[i]);
        memcpy(pa.data() + h_poff[i], hdb.sa.data() + hdb.off[i], h_len[i]);
    }
    d_s3.reserve(tot + 64);
    d_sa.reserve(tot + 64);
    d_off.reserve((size_t)n + 1);
    d_len.reserve(std::max<size_t>(n, 1));
    UC_HIP(hipMemcpy(d_s3.p, p3.data(), tot + 64, hipMemcpyHostToDevice));
    UC_HIP(hipMemcpy(d_sa.p, pa.data(), tot + 64, hipMemcpyHostToDevice));
    UC_HIP(hipMemcpy(d_off.p, h_poff.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
    if (n) UC_HIP(hipMemcpy(d_len.p, h_len.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    ddb.n = n;
    ddb.s3 = d_s3.p; ddb.sa = d_sa.p; ddb.off = d_off.p; ddb.len = d_len.p;
    ddb.S3 = d_S3.p; ddb.SA = d_SA.p;
    have_db = true;
    hit_cnt.assign(n, 0);
    hit_off.assign((size_t)n + 1, 0);
    hits.clear(); alns.clear(); edges.clear();
    aln_done.assign(n, 0);
    stats.n_seqs = n;
    stats.n_residues = hdb.residues();
    stats.algorithmic_bytes[UC_ST_LOAD] += 2 * hdb.residues() + 12ull * n;
    stats.stage_seconds[UC_ST_LOAD] += tm.seconds();
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
    launch_ungapped(ddb, n, dq.p, dt.p, dd.p, ds.p, stream);
    UC_HIP(hipGetLastError());
    UC_HIP(hipMemcpyAsync(out, ds.p, n * 4, hipMemcpyDeviceToHost, stream));
    UC_HIP(hipStreamSynchronize(stream));
}

// Gapped DP for a list of pairs (any order).  Host side: group by query, pick the (G,R) class from the
// query length, sort each query's targets by length so the alignments sharing a wave finish together,
// cut into workgroup tasks; device side: one launch per class (uc_sw_impl.hpp).
void Engine::sw_batch(int mode, const std::vector<PairIn> &pairs, int32_t *score, int32_t *qe, int32_t *te) {
    if (!have_db) fail(UC_ERR_ARGS, "no database loaded");
    const size_t n = pairs.size();
    if (n == 0) return;
    UC_HIP(hipSetDevice(device));
    const bool track = mode != 1;
    for (size_t i = 0; i < n; i++) {
        const PairIn &x = pairs[i];
        if (x.q >= hdb.n || x.t >= hdb.n) fail(UC_ERR_ARGS, "sw_batch: sequence id out of range");
        if (mode == 2 && (x.qe < 0 || x.te < 0 || (uint32_t)x.qe >= h_len[x.q] || (uint32_t)x.te >= h_len[x.t]))
            fail(UC_ERR_ARGS, "sw_batch: end position out of range");
    }
    // ---- order: by query (stable), then effective target length descending
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    bool grouped = true;
    for (size_t i = 1; i < n && grouped; i++) grouped = pairs[i].q >= pairs[i - 1].q;
    auto tlen = [&](uint32_t i) { return mode == 2 ? (uint32_t)pairs[i].te + 1 : h_len[pairs[i].t]; };
    if (!grouped) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return pairs[a].q < pairs[b].q; });
    struct Task { SwTask t; uint64_t work; };
    constexpr int NCLS = 17;   // 16 group classes + generic
    static const int cls_g[16] = {16, 16, 16, 16, 16, 16, 16, 16, 32, 32, 32, 32, 64, 64, 64, 64};
    static const int cls_r[16] = {4, 8, 12, 16, 20, 24, 28, 32, 20, 24, 28, 32, 20, 24, 28, 32};
    std::vector<std::vector<uint32_t>> cls_pairs(NCLS);   // sorted pair order per class
    std::vector<std::vector<Task>> cls_tasks(NCLS);
    uint64_t alg_bytes = 0;
    for (size_t b = 0; b < n;) {
        size_t e = b;
        const uint32_t q = pairs[order[b]].q;
        while (e < n && pairs[order[e]].q == q) e++;
        std::sort(order.begin() + b, order.begin() + e, [&](uint32_t x, uint32_t y) {
            uint32_t lx = tlen(x), ly = tlen(y);
            return lx != ly ? lx > ly : x < y;
        });
        int G, R, c = 16;
        if (sw_class_for((int)h_len[q], &G, &R))
            for (c = 0; c < 16; c++) if (cls_g[c] == G && cls_r[c] == R) break;
        std::vector<uint32_t> &cp = cls_pairs[c];
        for (size_t k = b; k < e; k += 256) {
            const size_t cnt = std::min<size_t>(256, e - k);
            uint64_t work = 0;
            for (size_t m = k; m < k + cnt; m++) work += tlen(order[m]);
            cls_tasks[c].push_back({{q, (uint32_t)cp.size(), (uint32_t)cnt}, work});
            for (size_t m = k; m < k + cnt; m++) {
                cp.push_back(order[m]);
                const uint32_t ql = mode == 2 ? (uint32_t)pairs[order[m]].qe + 1 : h_len[q];
                alg_bytes += 2ull * (ql + tlen(order[m])) + 32;
            }
        }
        b = e;
    }
    // ---- flatten: pairs of all classes back to back
    std::vector<uint32_t> flat; flat.reserve(n);
    std::vector<uint32_t> cls_base(NCLS + 1, 0);
    for (int c = 0; c < NCLS; c++) { cls_base[c] = (uint32_t)flat.size(); flat.insert(flat.end(), cls_pairs[c].begin(), cls_pairs[c].end()); }
    cls_base[NCLS] = (uint32_t)flat.size();
    std::vector<uint32_t> h_pt(n), h_pq;
    std::vector<int32_t> h_qe, h_te;
    for (size_t i = 0; i < n; i++) h_pt[i] = pairs[flat[i]].t;
    if (mode == 2) {
        h_qe.resize(n); h_te.resize(n);
        for (size_t i = 0; i < n; i++) { h_qe[i] = pairs[flat[i]].qe; h_te[i] = pairs[flat[i]].te; }
    }
    std::vector<SwTask> h_tasks;
    std::vector<uint32_t> task_base(NCLS + 1, 0);
    for (int c = 0; c < 16; c++) {
        task_base[c] = (uint32_t)h_tasks.size();
        std::vector<Task> &tv = cls_tasks[c];
        std::stable_sort(tv.begin(), tv.end(), [](const Task &a, const Task &b) { return a.work > b.work; });   // big first
        for (Task &t : tv) { t.t.begin += cls_base[c]; h_tasks.push_back(t.t); }
    }
    task_base[16] = task_base[NCLS] = (uint32_t)h_tasks.size();

    DevBuf<uint32_t> d_pt, d_pq;
    DevBuf<int32_t> d_qe, d_te, d_os, d_oq, d_ot, d_work;
    DevBuf<SwTask> d_tasks;
    d_pt.reserve(n); d_os.reserve(n);
    UC_HIP(hipMemcpyAsync(d_pt.p, h_pt.data(), n * 4, hipMemcpyHostToDevice, stream));
    if (mode == 2) {
        d_qe.reserve(n); d_te.reserve(n);
        UC_HIP(hipMemcpyAsync(d_qe.p, h_qe.data(), n * 4, hipMemcpyHostToDevice, stream));
        UC_HIP(hipMemcpyAsync(d_te.p, h_te.data(), n * 4, hipMemcpyHostToDevice, stream));
    }
    if (track) { d_oq.reserve(n); d_ot.reserve(n); }
    if (!h_tasks.empty()) {
        d_tasks.reserve(h_tasks.size());
        UC_HIP(hipMemcpyAsync(d_tasks.p, h_tasks.data(), h_tasks.size() * sizeof(SwTask), hipMemcpyHostToDevice, stream));
    }
    SwArgs a;
    a.db = ddb; a.tasks = d_tasks.p; a.pt = d_pt.p; a.pqe = d_qe.p; a.pte = d_te.p;
    a.oscore = d_os.p; a.oqe = d_oq.p; a.ote = d_ot.p; a.open = p.gap_open; a.ext = p.gap_ext;

    uint64_t launches = 0;
    timed_ms_begin();
    for (int c = 0; c < 16; c++) {
        const uint32_t nt = task_base[c + 1] - task_base[c];
        if (!nt) continue;
        SwArgs ac = a;
        ac.tasks = d_tasks.p + task_base[c];
        launch_sw_class(cls_g[c], cls_r[c], mode, ac, nt, stream);
        UC_HIP(hipGetLastError());
        launches++;
    }
    const uint32_t ngen = cls_base[17] - cls_base[16];
    if (ngen) {   // long queries: generic one-lane-per-pair kernel
        h_pq.resize(ngen);
        uint32_t max_lq = 1;
        for (uint32_t i = 0; i < ngen; i++) { h_pq[i] = pairs[flat[cls_base[16] + i]].q; max_lq = std::max(max_lq, h_len[h_pq[i]]); }
        d_pq.reserve(ngen);
        UC_HIP(hipMemcpyAsync(d_pq.p, h_pq.data(), (size_t)ngen * 4, hipMemcpyHostToDevice, stream));
        const size_t lanes = (size_t)((ngen + 63) / 64) * 64;
        d_work.reserve(2 * (size_t)max_lq * lanes);
        SwArgs ag = a;
        ag.pt = d_pt.p + cls_base[16];
        ag.pqe = mode == 2 ? d_qe.p + cls_base[16] : nullptr;
        ag.pte = mode == 2 ? d_te.p + cls_base[16] : nullptr;
        ag.oscore = d_os.p + cls_base[16];
        ag.oqe = track ? d_oq.p + cls_base[16] : nullptr;
        ag.ote = track ? d_ot.p + cls_base[16] : nullptr;
        launch_sw_generic(mode, ag, ngen, d_pq.p, d_work.p, max_lq, stream);
        UC_HIP(hipGetLastError());
        launches++;
    }
    const double ms = timed_ms_end();
    stats.sw_kernel_ms += ms;
    stats.sw_kernel_launches += launches;
    stats.sw_algorithmic_bytes += alg_bytes;

    std::vector<int32_t> r_s(n), r_q, r_t;
    UC_HIP(hipMemcpyAsync(r_s.data(), d_os.p, n * 4, hipMemcpyDeviceToHost, stream));
    if (track) {
        r_q.resize(n); r_t.resize(n);
        UC_HIP(hipMemcpyAsync(r_q.data(), d_oq.p, n * 4, hipMemcpyDeviceToHost, stream));
        UC_HIP(hipMemcpyAsync(r_t.data(), d_ot.p, n * 4, hipMemcpyDeviceToHost, stream));
    }
    UC_HIP(hipStreamSynchronize(stream));
    for (size_t i = 0; i < n; i++) {
        score[flat[i]] = r_s[i];
        if (track && qe) qe[flat[i]] = r_q[i];
        if (track && te) te[flat[i]] = r_t[i];
    }
}

void Engine::set_hits(const uint32_t *counts, const uc_hit *h) {
    const uint32_t n = hdb.n;
    hit_cnt.assign(counts, counts + n);
    hit_off.assign((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (counts[i] > (uint32_t)p.max_seqs) fail(UC_ERR_ARGS, "hit list of query %u longer than max_seqs", i);
        hit_off[i + 1] = hit_off[i] + counts[i];
    }
    hits.assign(h, h + hit_off[n]);
    for (const uc_hit &x : hits) if (x.target >= n) fail(UC_ERR_ARGS, "hit target out of range");
    alns.assign(hits.size(), uc_aln{});
    aln_done.assign(n, 0);
    edges.clear();
}

// Stage E5 + E6 for queries [qbegin, qend): forward pass (score + end), reversed-query pass, E-value
// gate on the corrected score, start pass for the survivors, coverage gate -> edges.
void Engine::align(uint32_t qbegin, uint32_t qend) {
    if (!have_db) fail(UC_ERR_ARGS, "no database loaded");
    if (qbegin > qend || qend > hdb.n) fail(UC_ERR_ARGS, "align: bad query range");
    if (p.min_seq_id > 0.0f)
        fail(UC_ERR_ARGS, "--min-seq-id > 0 needs the traceback pass, which this build does not implement yet");
    Timer tm;
    const uint64_t dbres = hdb.residues();
    const size_t CHUNK = 8u << 20;   // pairs per device batch
    for (uint32_t qa = qbegin; qa < qend;) {
        uint32_t qb = qa;
        while (qb < qend && (qb == qa || hit_off[qb + 1] - hit_off[qa] <= CHUNK)) qb++;
        const uint64_t b = hit_off[qa], e = hit_off[qb];
        const size_t n = (size_t)(e - b);
        if (n) {
            std::vector<PairIn> pairs(n);
            for (uint32_t q = qa; q < qb; q++)
                for (uint64_t k = hit_off[q]; k < hit_off[q + 1]; k++) pairs[k - b] = {q, hits[k].target, 0, 0};
            std::vector<int32_t> s0(n), q0(n), t0(n), s1(n, 0);
            sw_batch(0, pairs, s0.data(), q0.data(), t0.data());
            if (p.rev_correction) sw_batch(1, pairs, s1.data(), nullptr, nullptr);
            std::vector<PairIn> pass;
            std::vector<size_t> pass_idx;
            for (uint32_t q = qa; q < qb; q++) {
                const int32_t ms = min_score_for(p, (int)h_len[q], dbres);
                for (uint64_t k = hit_off[q]; k < hit_off[q + 1]; k++) {
                    const size_t i = (size_t)(k - b);
                    uc_aln &a = alns[k];
                    a = uc_aln{};
                    a.score = s0[i]; a.score_rev = s1[i]; a.corrected = s0[i] - s1[i];
                    a.qend = q0[i]; a.tend = t0[i]; a.qstart = -1; a.tstart = -1;
                    a.pass_evalue = (a.score > 0 && a.corrected >= ms);
                    stats.cells_fwd += (uint64_t)h_len[q] * h_len[pairs[i].t];
                    if (p.rev_correction) stats.cells_rev += (uint64_t)h_len[q] * h_len[pairs[i].t];
                    if (a.pass_evalue) {
                        pass.push_back({q, pairs[i].t, a.qend, a.tend});
                        pass_idx.push_back((size_t)k);
                        stats.cells_start += (uint64_t)(a.qend + 1) * (uint64_t)(a.tend + 1);
                    }
                }
            }
            std::vector<int32_t> s2(pass.size()), q2(pass.size()), t2(pass.size());
            sw_batch(2, pass, s2.data(), q2.data(), t2.data());
            for (size_t m = 0; m < pass.size(); m++) {
                uc_aln &a = alns[pass_idx[m]];
                if (s2[m] != a.score)
                    fail(UC_ERR_GENERIC, "start pass score %d != forward score %d (query %u target %u)", s2[m], a.score, pass[m].q, pass[m].t);
                a.qstart = a.qend - q2[m];
                a.tstart = a.tend - t2[m];
                const float qcov = (float)(a.qend - a.qstart + 1) / (float)h_len[pass[m].q];
                const float tcov = (float)(a.tend - a.tstart + 1) / (float)h_len[pass[m].t];
                const bool ok = p.cov_mode == 0 ? (qcov >= p.cov && tcov >= p.cov) : p.cov_mode == 1 ? (tcov >= p.cov) : (qcov >= p.cov);
                a.accepted = ok;
                if (ok) { edges.push_back(pass[m].q); edges.push_back(pass[m].t); }
            }
            stats.n_gapped_alignments += n;
            stats.n_start_alignments += pass.size();
        }
        for (uint32_t q = qa; q < qb; q++) aln_done[q] = 1;
        qa = qb;
    }
    stats.n_edges = edges.size() / 2;
    stats.algorithmic_bytes[UC_ST_GAPPED] = stats.sw_algorithmic_bytes;
    stats.stage_seconds[UC_ST_GAPPED] += tm.seconds();
}

// host merge of per-shard hit lists (multi-GPU exchange): per query keep the top
// max_seqs under the frozen order (score desc, target asc)
void merge_hits(uint32_t n, int max_seqs, int n_parts, const uint32_t *const *counts, const uc_hit *const *hits,
                std::vector<uint32_t> &out_cnt, std::vector<uc_hit> &out_hits) {
    out_cnt.assign(n, 0);
    out_hits.clear();
    std::vector<uint64_t> pos((size_t)n_parts, 0);
    std::vector<uc_hit> tmp;
    for (uint32_t q = 0; q < n; q++) {
        tmp.clear();
        for (int s = 0; s < n_parts; s++) {
            tmp.insert(tmp.end(), hits[s] + pos[s], hits[s] + pos[s] + counts[s][q]);
            pos[s] += counts[s][q];
        }
        std::sort(tmp.begin(), tmp.end(), [](const uc_hit &a, const uc_hit &b) {
            return a.score != b.score ? a.score > b.score : a.target < b.target;
        });
        for (size_t i = 1; i < tmp.size(); i++)
            if (tmp[i].target == tmp[i - 1].target && tmp[i].score == tmp[i - 1].score)
                fail(UC_ERR_ARGS, "merge_hits: target %u appears in two shards for query %u", tmp[i].target, q);
        if (tmp.size() > (size_t)max_seqs) tmp.resize((size_t)max_seqs);
        out_cnt[q] = (uint32_t)tmp.size();
        out_hits.insert(out_hits.end(), tmp.begin(), tmp.end());
    }
}

}  // namespace uc
