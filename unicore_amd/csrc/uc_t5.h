// uc_t5.h — ProstT5 AA -> 3Di encoder: model container, GGUF I/O, forward pass, createdb.
// Reference call site: /root/reference/src/modules/createdb.rs:157-166 (`foldseek createdb <fasta> <db> --prostt5-model
// <dir> [--gpu 1]`), weights file `<dir>/prostt5-f16.gguf` (createdb.rs:148).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace uc {

struct T5Config {
    int vocab = 150, d_model = 1024, d_kv = 128, n_heads = 32, d_ff = 16384, n_layers = 24;
    int rel_buckets = 32, rel_max_dist = 128;
    float eps = 1e-6f;
    int cnn_hidden = 32, cnn_kernel = 7, n_out = 20;
    int prefix_token = 149, eos_token = 1, unk_token = 2;      // "<AA2fold>", "</s>", "<unk>"
    // head convention — EXT-UNVERIFIED for Foldseek, so the DEFAULT is the reading this library has produced since its first createdb (ADVICE r3:
    // do not move every 3Di database on unverified grounds): </s>'s final hidden state feeds the CNN like any other position, and B/O/U/Z are
    // looked up in the vocabulary like every other letter (the GGUF vocabulary holds them; only letters it lacks fall back to X).
    // UC_T5_EOS_IN_HEAD=0 / UC_T5_KEEP_UZOB=0 select the other reading — ProstT5's published predict_3Di script run on one sequence: </s> is
    // attended by the encoder but masked to zero before the CNN, U/Z/O/B are read as X.  Both are tested against the oracle at full depth and
    // both have committed fixtures; INTEGRATION.md section D names what to diff once a Foldseek binary is at hand.
    int eos_in_head = 1, uzob_to_x = 0;
};

struct T5AttnTile { int32_t tok0, len, q0; };                  // sequence start token, its length, first query row of the workgroup (128 rows)

// ---- GGUF (v3) container: what llama.cpp / ggml and Foldseek's ProstT5 weights use --------------------------------------
struct GgufTensor {
    std::string name;
    std::vector<uint64_t> ne;        // ggml order: ne[0] is the contiguous dimension
    uint32_t type = 0;               // 0 = F32, 1 = F16 (the only types this loader accepts)
    uint64_t offset = 0;             // relative to the data section
    uint64_t n_elems() const { uint64_t n = 1; for (uint64_t d : ne) n *= d; return n; }
};
struct GgufFile {
    std::map<std::string, std::string> kv_str;
    std::map<std::string, double> kv_num;
    std::map<std::string, std::vector<std::string>> kv_strarr;
    std::vector<GgufTensor> tensors;
    uint64_t data_offset = 0;
    std::string path;
    const GgufTensor *find(const std::string &name) const;
};
void gguf_read_header(const std::string &path, GgufFile &g);

// ---- kernels (uc_t5_kernels.hip) -------------------------------------------------------------------------------------------
void t5_gemm(int epi, const void *A, const void *W, void *out, int M, int N, int K, hipStream_t s);
void t5_embed(const int32_t *tok, const void *emb, float *hidden, int T, int D, int vocab, hipStream_t s);
void t5_rmsnorm(const float *x, const float *w, void *y, int T, int D, float eps, hipStream_t s);
void t5_attention(const void *qkv, const T5AttnTile *tiles, int n_tiles, const float *bias, int bias_span, int H, void *out, hipStream_t s);
void t5_cnn_head(const void *y, int ldy, const int32_t *seq_of, const int32_t *seq_off, const float *b1, const float *w2, const float *b2, float *h1,
                 uint8_t *codes, float *logits, int T, int C1, int KW, int NO, int eos_in_head, hipStream_t s);
void t5_f32_to_f16(const float *x, void *y, size_t n, hipStream_t s);

// ---- the model on one GPU ------------------------------------------------------------------------------------------------------
struct T5Stats {
    uint64_t n_seqs = 0, n_tokens = 0;
    double flops = 0;                // algorithmic FLOPs of the GEMMs + attention of everything encoded so far
    double gemm_ms = 0, attn_ms = 0, other_ms = 0, total_ms = 0;   // HIP-event times
};

struct T5Model {
    T5Config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    // weights (device): f16 matrices, fp32 norms / biases / small conv
    void *emb = nullptr;                                   // [vocab, d_model] f16
    struct Layer { void *wqkv, *wo, *wi, *wo2; float *attn_norm, *ffn_norm; };
    std::vector<Layer> layers;
    float *final_norm = nullptr;
    float *rel_bias = nullptr;                             // [n_heads][rel_buckets] fp32 (layer 0's table, shared by all layers as in T5)
    void *w_conv1 = nullptr;                               // [ldc1 = round_up(kernel * hidden, 128)][d_model] f16, row k * hidden + c
    float *b_conv1 = nullptr, *w_conv2 = nullptr, *b_conv2 = nullptr;
    int ldc1 = 0;
    int aa_token[256];                                     // ASCII letter -> token id
    std::vector<void *> allocs;
    // activations (grown on demand)
    size_t cap_tokens = 0;
    float *hidden = nullptr, *h1 = nullptr, *logits = nullptr, *bias_tab = nullptr;
    void *xn = nullptr, *qkv = nullptr, *ao = nullptr, *ff = nullptr, *ycnn = nullptr;
    int32_t *d_tok = nullptr, *d_seq_of = nullptr, *d_seq_off = nullptr;
    T5AttnTile *d_tiles = nullptr;
    uint8_t *d_codes = nullptr;
    size_t cap_seqs = 0, cap_tiles = 0;
    int bias_span = 0;
    T5Stats stats;

    T5Model() = default;
    T5Model(const T5Model &) = delete;
    T5Model &operator=(const T5Model &) = delete;
    ~T5Model();
    void load(const std::string &gguf_path, int device);
    // encode a batch: seqs[i] = residue letters; out_codes[i] = 3Di states 0..19 per residue; optional logits (n_out per residue)
    void encode(const std::vector<std::string> &seqs, std::vector<std::vector<uint8_t>> &out_codes, std::vector<std::vector<float>> *out_logits = nullptr);
    // one batch of the plan: the sequences seqs[ids[k]] -> out_codes[ids[k]] (the slots of other sequences are not touched)
    void encode_ids(const std::vector<std::string> &seqs, const std::vector<uint32_t> &ids, std::vector<std::vector<uint8_t>> &out_codes,
                    std::vector<std::vector<float>> *out_logits);
    void encode_batch(const std::vector<const std::string *> &seqs, std::vector<std::vector<uint8_t>> &out_codes, size_t out_base,
                      std::vector<std::vector<float>> *out_logits);
};

// `foldseek createdb <fasta...> <out_db> --prostt5-model <dir>`: FASTA -> <db>, <db>_h, <db>_ss (predicted 3Di), .index, .dbtype, .lookup
// `devices`: one encoder replica (host thread + weights + activations) per entry; the sequences are sharded over the replicas, no collective (uc_t5.cpp)
void t5_createdb(const std::vector<std::string> &fasta_paths, const std::string &out_db, const std::string &model_path, const std::vector<int> &devices, int verbosity,
                 T5Stats *stats_out, std::vector<T5Stats> *per_replica = nullptr);
std::vector<std::vector<uint32_t>> t5_plan_batches(const std::vector<std::string> &seqs);      // length-sorted, token-bounded batches (ids into seqs), longest first
void t5_encode_replicated(const std::vector<std::string> &seqs, const std::string &gguf, const std::vector<int> &devices,
                          std::vector<std::vector<uint8_t>> &out_codes, T5Config *cfg_out, T5Stats *stats_out, std::vector<T5Stats> *per_replica);

}  // namespace uc
