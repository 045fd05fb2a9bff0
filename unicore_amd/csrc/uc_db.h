// uc_db.h — MMseqs/Foldseek-style database files on the boundary of the hot path.
// Format witness in the reference: src/seq/create_gene_specific_fasta.rs:9-36 (entries "TEXT\n\0",
// <db>/<db>_ss/<db>_h index-aligned); SURVEY.md Appendix B for .index/.dbtype/.lookup.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace uc {

struct HostDb {
    uint32_t n = 0;
    std::vector<uint64_t> keys;     // DB keys in ascending order; internal id = rank
    std::vector<uint64_t> off;      // n+1, byte offsets into s3/sa (no padding)
    std::vector<uint8_t> s3, sa;    // codes 0..20
    std::vector<std::string> names; // first token of the header entry (may be empty if no _h)
    uint64_t residues() const { return off.empty() ? 0 : off.back(); }
    uint32_t len(uint32_t i) const { return (uint32_t)(off[i + 1] - off[i]); }
};

struct IndexEntry { uint64_t key, off, len; };
std::vector<IndexEntry> read_index(const std::string &path);   // sorted by key
std::string read_whole_file(const std::string &path);

// read <prefix> (AA), <prefix>_ss (3Di), and <prefix>_h if with_headers
void read_seq_db(const std::string &prefix, HostDb &db, bool with_headers);

// cluster DB (== foldseek cluster output kept by `-k`, cluster.rs:43,67-76): one entry per representative,
// member keys one per line (representative first, then ascending), "\0"-terminated; dbtype 6
void write_cluster_db(const std::string &prefix, const std::vector<uint64_t> &keys, const uint32_t *assign, uint32_t n);
// == foldseek createtsv (cluster.rs:59-64): "rep_name\tmember_name\n"
void create_tsv(const std::string &db_prefix, const std::string &cluster_db, const std::string &out_tsv);
// alignment DB (== `foldseek search` result kept by `unicore search -k`, search.rs:44-61): one entry per query key;
// rows "targetKey bits fident evalue qstart qend qlen tstart tend tlen alnlen idents gapopen corrected" (tab separated,
// positions 0-based, MMseqs2 column order + the integer statistics convertalis needs), entry "\0"-terminated; dbtype 5
struct AlnRow {
    uint64_t tkey;
    int32_t bits, qstart, qend, qlen, tstart, tend, tlen, aln_len, idents, gap_opens, corrected;
    double fident, evalue;
};
void write_aln_db(const std::string &prefix, const std::vector<uint64_t> &qkeys, const std::vector<std::vector<AlnRow>> &rows);
// == foldseek convertalis (search.rs:52-57): BLAST-tab "query target fident alnlen mismatch gapopen qstart qend tstart
// tend evalue bits", positions 1-based; names = first token of the header entries of the two DBs
void convert_alis(const std::string &query_db, const std::string &target_db, const std::string &aln_db, const std::string &out_m8);
// == foldseek rmdb (cluster.rs:67-76)
void remove_db(const std::string &prefix);

}  // namespace uc
