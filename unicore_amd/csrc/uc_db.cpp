// uc_db.cpp — DB reader / cluster-DB writer / createtsv / rmdb (host side of the boundary).
#include "uc_db.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <thread>
#include <unordered_map>

#include "uc_common.h"
#include "uc_options.h"

namespace uc {

std::string read_whole_file(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) fail(UC_ERR_IO, "cannot open %s", path.c_str());
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::string s;
    s.resize((size_t)n);
    if (n > 0 && fread(&s[0], 1, (size_t)n, f) != (size_t)n) { fclose(f); fail(UC_ERR_IO, "short read on %s", path.c_str()); }
    fclose(f);
    return s;
}

std::vector<IndexEntry> read_index(const std::string &path) {
    std::string txt = read_whole_file(path);
    std::vector<IndexEntry> v;
    const char *s = txt.c_str(), *end = s + txt.size();
    auto num = [&](uint64_t &out) -> bool {
        while (s < end && (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\n')) s++;
        if (s >= end || *s < '0' || *s > '9') return false;
        uint64_t x = 0;
        while (s < end && *s >= '0' && *s <= '9') x = x * 10 + (uint64_t)(*s++ - '0');
        out = x;
        return true;
    };
    for (;;) {
        IndexEntry e;
        if (!num(e.key)) break;
        if (!num(e.off) || !num(e.len)) fail(UC_ERR_IO, "malformed index line in %s", path.c_str());
        v.push_back(e);
    }
    std::sort(v.begin(), v.end(), [](const IndexEntry &a, const IndexEntry &b) { return a.key < b.key; });
    return v;
}

namespace {
// read-only view of a whole file (mmap): the two data files of a sequence DB are only read once, letter by letter
struct MappedFile {
    const char *p = nullptr;
    size_t n = 0;
    explicit MappedFile(const std::string &path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) fail(UC_ERR_IO, "cannot open %s", path.c_str());
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); fail(UC_ERR_IO, "cannot stat %s", path.c_str()); }
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); fail(UC_ERR_IO, "cannot map %s", path.c_str()); }
            p = (const char *)m;
        }
        close(fd);
    }
    ~MappedFile() { if (p) munmap((void *)p, n); }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    const char *data() const { return p; }
    size_t size() const { return n; }
};
}  // namespace

void read_seq_db(const std::string &prefix, HostDb &db, bool with_headers) {
    std::vector<IndexEntry> ia, is;
    {   // the two index files are parsed side by side (errors are re-thrown on this thread)
        std::exception_ptr err;
        std::thread t([&] { try { is = read_index(prefix + "_ss.index"); } catch (...) { err = std::current_exception(); } });
        try { ia = read_index(prefix + ".index"); } catch (...) { t.join(); throw; }
        t.join();
        if (err) std::rethrow_exception(err);
    }
    if (ia.size() != is.size()) fail(UC_ERR_IO, "%s and %s_ss have different entry counts", prefix.c_str(), prefix.c_str());
    if (ia.size() >= (1u << 24)) fail(UC_ERR_ARGS, "database has %zu sequences; this build supports < 2^24", ia.size());
    const MappedFile da(prefix), ds(prefix + "_ss");
    db.n = (uint32_t)ia.size();
    db.keys.resize(db.n);
    db.off.assign((size_t)db.n + 1, 0);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < db.n; i++) {
        if (ia[i].key != is[i].key) fail(UC_ERR_IO, "key mismatch between AA and 3Di index at entry %u", i);
        uint64_t l = ia[i].len >= 2 ? ia[i].len - 2 : 0, l2 = is[i].len >= 2 ? is[i].len - 2 : 0;
        if (l != l2) fail(UC_ERR_IO, "AA/3Di length mismatch for key %llu", (unsigned long long)ia[i].key);
        if (ia[i].off + l > da.size() || is[i].off + l > ds.size()) fail(UC_ERR_IO, "index entry beyond data file (key %llu)", (unsigned long long)ia[i].key);
        if (l > 65535) fail(UC_ERR_ARGS, "sequence key %llu longer than 65535 (max-seq-len)", (unsigned long long)ia[i].key);
        db.keys[i] = ia[i].key;
        db.off[i] = tot;
        tot += l;
    }
    db.off[db.n] = tot;
    db.s3.resize(tot);
    db.sa.resize(tot);
    // letters -> codes: table lookup, sequences split over a few threads (the single-threaded switch per letter was most
    // of the 0.2 s this function took for 2 x 47 MB)
    uint8_t lut[256];
    for (int c = 0; c < 256; c++) lut[c] = (uint8_t)letter_code((char)c);
    auto encode = [&](uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; i++) {
            const uint64_t l = db.off[i + 1] - db.off[i];
            const unsigned char *pa = (const unsigned char *)da.data() + ia[i].off, *ps = (const unsigned char *)ds.data() + is[i].off;
            uint8_t *oa = db.sa.data() + db.off[i], *os = db.s3.data() + db.off[i];
            for (uint64_t k = 0; k < l; k++) { oa[k] = lut[pa[k]]; os[k] = lut[ps[k]]; }
        }
    };
    const unsigned T = tot < (1u << 22) ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (T == 1) {
        encode(0, db.n);
    } else {
        std::vector<std::thread> th;
        uint32_t b = 0;
        for (unsigned t = 0; t < T; t++) {           // contiguous sequence ranges with ~equal residue counts
            uint32_t e = t + 1 == T ? db.n : b;
            while (e < db.n && db.off[e] < tot * (t + 1) / T) e++;
            th.emplace_back(encode, b, e);
            b = e;
        }
        for (auto &x : th) x.join();
    }
    db.names.clear();
    if (with_headers) {
        std::vector<IndexEntry> ih = read_index(prefix + "_h.index");
        if (ih.size() != ia.size()) fail(UC_ERR_IO, "%s_h has a different entry count", prefix.c_str());
        std::string dh = read_whole_file(prefix + "_h");
        db.names.resize(db.n);
        for (uint32_t i = 0; i < db.n; i++) {
            if (ih[i].key != ia[i].key) fail(UC_ERR_IO, "key mismatch between header and sequence index at entry %u", i);
            size_t b = ih[i].off, e = b;
            while (e < dh.size() && dh[e] && dh[e] != ' ' && dh[e] != '\t' && dh[e] != '\n') e++;
            db.names[i] = dh.substr(b, e - b);
        }
    }
}

static void write_dbtype(const std::string &path, int32_t t) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) fail(UC_ERR_IO, "cannot write %s", path.c_str());
    fwrite(&t, 4, 1, f);
    fclose(f);
}

void write_cluster_db(const std::string &prefix, const std::vector<uint64_t> &keys, const uint32_t *assign, uint32_t n) {
    std::vector<uint64_t> cnt((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (assign[i] >= n) fail(UC_ERR_GENERIC, "invalid cluster assignment for sequence %u", i);
        cnt[assign[i] + 1]++;
    }
    for (uint32_t i = 0; i < n; i++) cnt[i + 1] += cnt[i];
    std::vector<uint32_t> mem(n);
    std::vector<uint64_t> cur(cnt.begin(), cnt.end() - 1);
    for (uint32_t i = 0; i < n; i++) mem[cur[assign[i]]++] = i;   // ascending id within each cluster
    std::string tmpd = prefix + ".tmp_data", tmpi = prefix + ".tmp_index";
    FILE *fd = fopen(tmpd.c_str(), "wb"), *fi = fopen(tmpi.c_str(), "wb");
    if (!fd || !fi) { if (fd) fclose(fd); if (fi) fclose(fi); fail(UC_ERR_IO, "cannot write cluster DB %s", prefix.c_str()); }
    uint64_t off = 0;
    char buf[32];
    for (uint32_t r = 0; r < n; r++) {
        if (cnt[r + 1] == cnt[r]) continue;
        if (assign[r] != r) fail(UC_ERR_GENERIC, "cluster %u has members but is not its own representative", r);
        uint64_t len = 0;
        int k = snprintf(buf, sizeof buf, "%llu\n", (unsigned long long)keys[r]);
        fwrite(buf, 1, (size_t)k, fd); len += (uint64_t)k;
        for (uint64_t m = cnt[r]; m < cnt[r + 1]; m++) {
            if (mem[m] == r) continue;
            k = snprintf(buf, sizeof buf, "%llu\n", (unsigned long long)keys[mem[m]]);
            fwrite(buf, 1, (size_t)k, fd); len += (uint64_t)k;
        }
        fputc(0, fd); len += 1;
        fprintf(fi, "%llu\t%llu\t%llu\n", (unsigned long long)keys[r], (unsigned long long)off, (unsigned long long)len);
        off += len;
    }
    bool bad = ferror(fd) || ferror(fi);
    bad |= fclose(fd) != 0;
    bad |= fclose(fi) != 0;
    if (bad) fail(UC_ERR_IO, "write error on cluster DB %s", prefix.c_str());
    // never leave a partial result behind with exit 0 (SURVEY.md 8b "errors")
    if (rename(tmpd.c_str(), prefix.c_str()) != 0 || rename(tmpi.c_str(), (prefix + ".index").c_str()) != 0)
        fail(UC_ERR_IO, "cannot finalize cluster DB %s", prefix.c_str());
    write_dbtype(prefix + ".dbtype", 6);
}

void create_tsv(const std::string &db_prefix, const std::string &cluster_db, const std::string &out_tsv) {
    // names: first whitespace-delimited token of each header entry (for Unicore DBs exactly
    // "unicore_<10 hex>", createdb.rs:104-106)
    std::vector<IndexEntry> ih = read_index(db_prefix + "_h.index");
    std::string dh = read_whole_file(db_prefix + "_h");
    std::unordered_map<uint64_t, std::string> name;
    name.reserve(ih.size() * 2);
    for (const IndexEntry &e : ih) {
        size_t b = e.off, x = b;
        while (x < dh.size() && dh[x] && dh[x] != ' ' && dh[x] != '\t' && dh[x] != '\n') x++;
        name[e.key] = dh.substr(b, x - b);
    }
    std::vector<IndexEntry> ic = read_index(cluster_db + ".index");
    std::string dc = read_whole_file(cluster_db);
    std::string tmp = out_tsv + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) fail(UC_ERR_IO, "cannot write %s", out_tsv.c_str());
    for (const IndexEntry &e : ic) {
        auto rit = name.find(e.key);
        if (rit == name.end()) { fclose(f); fail(UC_ERR_IO, "cluster representative key %llu not in %s_h", (unsigned long long)e.key, db_prefix.c_str()); }
        size_t p = e.off, end = std::min<size_t>(e.off + e.len, dc.size());
        while (p < end && dc[p]) {
            uint64_t k = 0; bool any = false;
            while (p < end && dc[p] >= '0' && dc[p] <= '9') { k = k * 10 + (uint64_t)(dc[p++] - '0'); any = true; }
            while (p < end && dc[p] && dc[p] != '\n') p++;   // ignore extra columns
            if (p < end && dc[p] == '\n') p++;
            if (!any) continue;
            auto mit = name.find(k);
            if (mit == name.end()) { fclose(f); fail(UC_ERR_IO, "cluster member key %llu not in %s_h", (unsigned long long)k, db_prefix.c_str()); }
            fputs(rit->second.c_str(), f); fputc('\t', f); fputs(mit->second.c_str(), f); fputc('\n', f);
        }
    }
    bool bad = ferror(f);
    bad |= fclose(f) != 0;
    if (bad || rename(tmp.c_str(), out_tsv.c_str()) != 0) fail(UC_ERR_IO, "write error on %s", out_tsv.c_str());
}

void write_aln_db(const std::string &prefix, const std::vector<uint64_t> &qkeys, const std::vector<std::vector<AlnRow>> &rows) {
    std::string tmpd = prefix + ".tmp_data", tmpi = prefix + ".tmp_index";
    FILE *fd = fopen(tmpd.c_str(), "wb"), *fi = fopen(tmpi.c_str(), "wb");
    if (!fd || !fi) { if (fd) fclose(fd); if (fi) fclose(fi); fail(UC_ERR_IO, "cannot write alignment DB %s", prefix.c_str()); }
    // records are formatted by a few threads (contiguous query ranges, one text buffer each) and written in order: the
    // 14-field printf per row was most of uc_search's host time
    const size_t nq = qkeys.size();
    size_t nrows = 0;
    for (const auto &v : rows) nrows += v.size();
    const unsigned T = nrows < 20000 ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::string> text(T);
    std::vector<std::vector<uint64_t>> lens(T);
    auto format = [&](unsigned t) {
        const size_t qb = nq * t / T, qe = nq * (t + 1) / T;
        std::string &out = text[t];
        lens[t].resize(qe - qb);
        char buf[256];
        for (size_t q = qb; q < qe; q++) {
            const size_t start = out.size();
            for (const AlnRow &r : rows[q]) {
                const int k = snprintf(buf, sizeof buf, "%llu\t%d\t%.3f\t%.3E\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", (unsigned long long)r.tkey, r.bits,
                                       r.fident, r.evalue, r.qstart, r.qend, r.qlen, r.tstart, r.tend, r.tlen, r.aln_len, r.idents, r.gap_opens, r.corrected);
                if (k > 0) out.append(buf, (size_t)std::min<int>(k, (int)sizeof buf - 1));
            }
            out.push_back('\0');
            lens[t][q - qb] = out.size() - start;
        }
    };
    if (T == 1) format(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back(format, t);
        for (auto &x : th) x.join();
    }
    uint64_t off = 0;
    for (unsigned t = 0; t < T; t++) {
        if (!text[t].empty()) fwrite(text[t].data(), 1, text[t].size(), fd);
        const size_t qb = nq * t / T;
        for (size_t k = 0; k < lens[t].size(); k++) {
            fprintf(fi, "%llu\t%llu\t%llu\n", (unsigned long long)qkeys[qb + k], (unsigned long long)off, (unsigned long long)lens[t][k]);
            off += lens[t][k];
        }
    }
    bool bad = ferror(fd) || ferror(fi);
    bad |= fclose(fd) != 0;
    bad |= fclose(fi) != 0;
    if (bad) fail(UC_ERR_IO, "write error on alignment DB %s", prefix.c_str());
    if (rename(tmpd.c_str(), prefix.c_str()) != 0 || rename(tmpi.c_str(), (prefix + ".index").c_str()) != 0)
        fail(UC_ERR_IO, "cannot finalize alignment DB %s", prefix.c_str());
    write_dbtype(prefix + ".dbtype", 5);
}

static std::unordered_map<uint64_t, std::string> header_names(const std::string &db_prefix) {
    std::vector<IndexEntry> ih = read_index(db_prefix + "_h.index");
    std::string dh = read_whole_file(db_prefix + "_h");
    std::unordered_map<uint64_t, std::string> name;
    name.reserve(ih.size() * 2);
    for (const IndexEntry &e : ih) {
        size_t b = e.off, x = b;
        while (x < dh.size() && dh[x] && dh[x] != ' ' && dh[x] != '\t' && dh[x] != '\n') x++;
        name[e.key] = dh.substr(b, x - b);
    }
    return name;
}

void convert_alis(const std::string &query_db, const std::string &target_db, const std::string &aln_db, const std::string &out_m8) {
    const auto qname = header_names(query_db);
    const auto tname_own = query_db == target_db ? std::unordered_map<uint64_t, std::string>() : header_names(target_db);
    const auto &tname = query_db == target_db ? qname : tname_own;
    std::vector<IndexEntry> ia = read_index(aln_db + ".index");
    std::string da = read_whole_file(aln_db);
    std::string tmp = out_m8 + ".tmp";
    // rows are converted by a few threads (contiguous ranges of query records, one text buffer each) and written in order
    const size_t nrec = ia.size();
    const unsigned T = da.size() < (1u << 20) ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::string> text(T), err(T);
    auto convert = [&](unsigned t) {
        std::string &out = text[t];
        char buf[512];
        for (size_t r = nrec * t / T; r < nrec * (t + 1) / T; r++) {
            const IndexEntry &e = ia[r];
            auto qit = qname.find(e.key);
            if (qit == qname.end()) { err[t] = "alignment DB query key " + std::to_string(e.key) + " not in " + query_db + "_h"; return; }
            size_t p = e.off;
            const size_t end = std::min<size_t>(e.off + e.len, da.size());
            while (p < end && da[p]) {
                size_t eol = p;
                while (eol < end && da[eol] && da[eol] != '\n') eol++;
                // 14 tab-separated fields; fields 2 and 3 (fident, evalue) are passed through as text
                const char *fld[14];
                size_t flen[14];
                int nf = 0;
                for (size_t b = p; nf < 14;) {
                    size_t x = b;
                    while (x < eol && da[x] != '\t') x++;
                    fld[nf] = da.data() + b; flen[nf] = x - b; nf++;
                    if (x >= eol) break;
                    b = x + 1;
                }
                if (nf != 14) { err[t] = "malformed row in alignment DB " + aln_db + ": '" + da.substr(p, eol - p) + "'"; return; }
                auto num = [&](int k) { return strtoll(fld[k], nullptr, 10); };
                const unsigned long long tkey = strtoull(fld[0], nullptr, 10);
                const int bits = (int)num(1), qs = (int)num(4), qe = (int)num(5), ts = (int)num(7), te = (int)num(8), alen = (int)num(10),
                          idents = (int)num(11), gaps = (int)num(12);
                p = eol < end && da[eol] == '\n' ? eol + 1 : eol;
                auto tit = tname.find(tkey);
                if (tit == tname.end()) { err[t] = "alignment DB target key " + std::to_string(tkey) + " not in " + target_db + "_h"; return; }
                const int pairs = (qe - qs + 1) + (te - ts + 1) - alen;
                out.append(qit->second).push_back('\t');
                out.append(tit->second);
                const int k = snprintf(buf, sizeof buf, "\t%.*s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%.*s\t%d\n", (int)std::min<size_t>(flen[2], 64), fld[2], alen,
                                       pairs - idents, gaps, qs + 1, qe + 1, ts + 1, te + 1, (int)std::min<size_t>(flen[3], 64), fld[3], bits);
                if (k > 0) out.append(buf, (size_t)std::min<int>(k, (int)sizeof buf - 1));
            }
        }
    };
    if (T == 1) convert(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back(convert, t);
        for (auto &x : th) x.join();
    }
    for (unsigned t = 0; t < T; t++)
        if (!err[t].empty()) fail(UC_ERR_IO, "%s", err[t].c_str());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) fail(UC_ERR_IO, "cannot write %s", out_m8.c_str());
    for (unsigned t = 0; t < T; t++)
        if (!text[t].empty()) fwrite(text[t].data(), 1, text[t].size(), f);
    bool bad = ferror(f);
    bad |= fclose(f) != 0;
    if (bad || rename(tmp.c_str(), out_m8.c_str()) != 0) fail(UC_ERR_IO, "write error on %s", out_m8.c_str());
}

void remove_db(const std::string &prefix) {
    static const char *sfx[] = {"", ".index", ".dbtype", ".lookup", ".source", "_h", "_h.index", "_h.dbtype", ".tmp_data", ".tmp_index"};
    for (const char *s : sfx) unlink((prefix + s).c_str());
}

}  // namespace uc
