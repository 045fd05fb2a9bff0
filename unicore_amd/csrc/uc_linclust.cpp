// uc_linclust.cpp — stage E8a: candidate pairs of the linear-time pre-clustering step (spec UC-1 E8a;
// restates Linclust, Steinegger & Soeding 2018 - the redundancy filter in front of Foldseek's default
// clustering workflow).  Host side: hashing 47 M k-mers and sorting 3 M kept entries is a fraction of a
// second; the alignments of the candidate pairs - the actual work - run through the GPU stages E5/E6 like any hit list.
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "uc_engine.h"

namespace uc {

namespace {
inline uint64_t lc_hash(uint32_t v) {   // SplitMix64 finaliser of the k-mer value
    uint64_t z = (uint64_t)v + 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}
struct Cand { uint64_t h; uint32_t v, pos; };
struct Ent { uint32_t v, seq; };
}  // namespace

std::vector<uint32_t> linclust_pairs(const HostDb &db, const Params &p, int threads) {
    const uint32_t n = db.n;
    const int m = p.kmer_per_seq;
    std::vector<std::vector<Ent>> part;
    const unsigned T = std::max(1, std::min(threads, 64));
    part.resize(T);
    auto work = [&](unsigned t) {   // every sequence keeps its m k-mers with the smallest (hash, position)
        std::vector<Cand> cand;
        const uint32_t lo = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T);
        for (uint32_t s = lo; s < hi; s++) {
            const uint8_t *x = db.s3.data() + db.off[s];
            const int64_t l = (int64_t)db.len(s);
            cand.clear();
            for (int64_t j = 0; j + p.span <= l && j <= 65535; j++) {
                uint32_t v = 0, mul = 1;
                bool ok = true;
                for (int k = 0; k < K; k++) {
                    const uint8_t c = x[j + p.koff[k]];
                    ok &= c < KA;
                    v += c * mul; mul *= KA;
                }
                if (ok) cand.push_back({lc_hash(v), v, (uint32_t)j});
            }
            const size_t keep = std::min<size_t>(cand.size(), (size_t)m);
            std::partial_sort(cand.begin(), cand.begin() + keep, cand.end(),
                              [](const Cand &a, const Cand &b) { return a.h != b.h ? a.h < b.h : a.pos < b.pos; });
            for (size_t k = 0; k < keep; k++) part[t].push_back({cand[k].v, s});
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    std::vector<Ent> ent;
    for (auto &v : part) ent.insert(ent.end(), v.begin(), v.end());
    std::sort(ent.begin(), ent.end(), [](const Ent &a, const Ent &b) { return a.v != b.v ? a.v < b.v : a.seq < b.seq; });
    std::vector<uint64_t> pr;   // (centre << 32 | member)
    for (size_t b = 0; b < ent.size();) {
        size_t e = b;
        while (e < ent.size() && ent[e].v == ent[b].v) e++;
        uint32_t c = ent[b].seq;               // centre: longest sequence of the group, ties: smallest id
        uint64_t lc = db.len(c);
        for (size_t k = b + 1; k < e; k++)
            if (db.len(ent[k].seq) > lc) { c = ent[k].seq; lc = db.len(c); }
        for (size_t k = b; k < e; k++)
            if (ent[k].seq != c && (k == b || ent[k].seq != ent[k - 1].seq)) pr.push_back(((uint64_t)c << 32) | ent[k].seq);
        b = e;
    }
    std::sort(pr.begin(), pr.end());
    pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
    std::vector<uint32_t> out(2 * pr.size());
    for (size_t k = 0; k < pr.size(); k++) { out[2 * k] = (uint32_t)(pr[k] >> 32); out[2 * k + 1] = (uint32_t)pr[k]; }
    return out;
}

}  // namespace uc
