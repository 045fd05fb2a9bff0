// uc_t5.cpp — host side of the ProstT5 AA -> 3Di encoder: GGUF reader, weight upload, batching, the layer loop, createdb.
// Kernels: uc_t5_kernels.hip.  Reference call site: /root/reference/src/modules/createdb.rs:137-166.
#include "uc_t5.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <atomic>
#include <numeric>
#include <thread>

#include "uc_common.h"
#include "uc_options.h"

namespace uc {

// ---------------------------------------------------------------------------------------------- GGUF
namespace {
struct Reader {
    const uint8_t *p, *end;
    template <typename T> T get() {
        if (p + sizeof(T) > end) fail(UC_ERR_IO, "GGUF: truncated file");
        T v; memcpy(&v, p, sizeof(T)); p += sizeof(T); return v;
    }
    std::string str() {
        const uint64_t n = get<uint64_t>();
        if (n > (uint64_t)(end - p)) fail(UC_ERR_IO, "GGUF: truncated string");
        std::string s((const char *)p, (size_t)n); p += n; return s;
    }
};
double read_scalar(Reader &r, uint32_t type) {
    switch (type) {
        case 0: return r.get<uint8_t>();  case 1: return r.get<int8_t>();   case 2: return r.get<uint16_t>(); case 3: return r.get<int16_t>();
        case 4: return r.get<uint32_t>(); case 5: return r.get<int32_t>();  case 6: return r.get<float>();    case 7: return r.get<uint8_t>();
        case 10: return (double)r.get<uint64_t>(); case 11: return (double)r.get<int64_t>(); case 12: return r.get<double>();
        default: fail(UC_ERR_IO, "GGUF: unsupported value type %u", type);
    }
}
struct Mapped {
    const uint8_t *p = nullptr; size_t n = 0;
    explicit Mapped(const std::string &path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) fail(UC_ERR_IO, "cannot open %s", path.c_str());
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); fail(UC_ERR_IO, "cannot stat %s", path.c_str()); }
        n = (size_t)st.st_size;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) fail(UC_ERR_IO, "cannot map %s", path.c_str());
        p = (const uint8_t *)m;
    }
    ~Mapped() { if (p) munmap((void *)p, n); }
};
}  // namespace

const GgufTensor *GgufFile::find(const std::string &name) const {
    for (const GgufTensor &t : tensors) if (t.name == name) return &t;
    return nullptr;
}

void gguf_read_header(const std::string &path, GgufFile &g) {
    Mapped m(path);
    Reader r{m.p, m.p + m.n};
    if (r.get<uint32_t>() != 0x46554747u) fail(UC_ERR_IO, "%s is not a GGUF file", path.c_str());
    const uint32_t version = r.get<uint32_t>();
    if (version < 2 || version > 3) fail(UC_ERR_IO, "%s: GGUF version %u not supported (2, 3)", path.c_str(), version);
    const uint64_t n_tensors = r.get<uint64_t>(), n_kv = r.get<uint64_t>();
    g = GgufFile();
    g.path = path;
    uint64_t alignment = 32;
    for (uint64_t i = 0; i < n_kv; i++) {
        const std::string key = r.str();
        const uint32_t type = r.get<uint32_t>();
        if (type == 8) g.kv_str[key] = r.str();
        else if (type == 9) {
            const uint32_t et = r.get<uint32_t>();
            const uint64_t cnt = r.get<uint64_t>();
            if (et == 8) { std::vector<std::string> &v = g.kv_strarr[key]; v.reserve((size_t)cnt); for (uint64_t k = 0; k < cnt; k++) v.push_back(r.str()); }
            else for (uint64_t k = 0; k < cnt; k++) (void)read_scalar(r, et);
        } else g.kv_num[key] = read_scalar(r, type);
    }
    if (g.kv_num.count("general.alignment")) {
        const double a = g.kv_num["general.alignment"];
        if (!(a >= 1 && a <= 65536) || ((uint64_t)a & ((uint64_t)a - 1))) fail(UC_ERR_IO, "%s: general.alignment %g is not a power of two in [1, 65536]", path.c_str(), a);
        alignment = (uint64_t)a;
    }
    if (n_tensors > (1u << 20)) fail(UC_ERR_IO, "%s: %llu tensors?", path.c_str(), (unsigned long long)n_tensors);
    for (uint64_t i = 0; i < n_tensors; i++) {
        GgufTensor t;
        t.name = r.str();
        const uint32_t nd = r.get<uint32_t>();
        if (nd > 4) fail(UC_ERR_IO, "GGUF: tensor %s has %u dimensions", t.name.c_str(), nd);
        uint64_t elems = 1;
        for (uint32_t d = 0; d < nd; d++) {
            const uint64_t e = r.get<uint64_t>();
            // zero dims and products beyond 2^40 elements are not weights of this model: reject before any size arithmetic can wrap
            if (e == 0 || e > (1ull << 40) || elems > (1ull << 40) / e) fail(UC_ERR_IO, "GGUF: tensor %s has an empty or absurd shape", t.name.c_str());
            elems *= e;
            t.ne.push_back(e);
        }
        t.type = r.get<uint32_t>();
        t.offset = r.get<uint64_t>();
        g.tensors.push_back(t);
    }
    const uint64_t pos = (uint64_t)(r.p - m.p);
    g.data_offset = (pos + alignment - 1) / alignment * alignment;
    for (const GgufTensor &t : g.tensors) {
        if (t.type > 1) fail(UC_ERR_IO, "GGUF: tensor %s has ggml type %u; this loader reads F32 and F16 (prostt5-f16.gguf)", t.name.c_str(), t.type);
        const uint64_t bytes = t.n_elems() * (t.type == 0 ? 4 : 2);     // <= 2^42: cannot wrap
        if (g.data_offset > m.n || t.offset > m.n - g.data_offset || bytes > m.n - g.data_offset - t.offset) fail(UC_ERR_IO, "GGUF: tensor %s reaches beyond the end of the file", t.name.c_str());
    }
}

// ---------------------------------------------------------------------------------------------- model
namespace {

float half_to_float(uint16_t h) {
    const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, f = h & 1023;
    uint32_t u;
    if (e == 0) {
        if (f == 0) u = s << 31;
        else { int ee = -1; uint32_t ff = f; do { ee++; ff <<= 1; } while (!(ff & 1024)); u = (s << 31) | ((uint32_t)(127 - 15 - ee) << 23) | ((ff & 1023) << 13); }
    } else if (e == 31) u = (s << 31) | 0x7f800000u | (f << 13);
    else u = (s << 31) | ((e + 112) << 23) | (f << 13);
    float o; memcpy(&o, &u, 4); return o;
}

const GgufTensor *find_any(const GgufFile &g, std::initializer_list<std::string> names, bool required = true) {
    for (const std::string &n : names) if (const GgufTensor *t = g.find(n)) return t;
    if (required) fail(UC_ERR_IO, "%s: tensor '%s' (or one of its aliases) is missing", g.path.c_str(), names.begin()->c_str());
    return nullptr;
}

// T5 relative position bucket, bidirectional — mirrors transformers' T5Attention._relative_position_bucket in float32
int rel_bucket(int rel /* key - query */, int num_buckets, int max_distance) {
    const int nb = num_buckets / 2;
    int ret = rel > 0 ? nb : 0;
    const int n = rel < 0 ? -rel : rel;
    const int max_exact = nb / 2;
    if (n < max_exact) return ret + n;
    const float ratio = ::logf((float)n / (float)max_exact) / (float)std::log((double)max_distance / (double)max_exact);
    int v = max_exact + (int)(ratio * (float)(nb - max_exact));
    return ret + std::min(v, nb - 1);
}

}  // namespace

T5Model::~T5Model() {
    (void)hipSetDevice(device);
    for (void *p : allocs) (void)hipFree(p);
    for (void *p : {(void *)hidden, (void *)h1, (void *)logits, (void *)bias_tab, xn, qkv, ao, ff, ycnn, (void *)d_tok, (void *)d_seq_of, (void *)d_seq_off,
                    (void *)d_tiles, (void *)d_codes})
        if (p) (void)hipFree(p);
    if (ev[0]) (void)hipEventDestroy(ev[0]);
    if (ev[1]) (void)hipEventDestroy(ev[1]);
    if (stream) (void)hipStreamDestroy(stream);
}

void T5Model::load(const std::string &gguf_path, int dev) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) fail(UC_ERR_DEVICE, "no HIP device available; the ProstT5 encoder has no CPU fallback");
    if (dev < 0) UC_HIP(hipGetDevice(&dev));
    if (dev >= ndev) fail(UC_ERR_DEVICE, "device %d requested but only %d visible", dev, ndev);
    device = dev;
    UC_HIP(hipSetDevice(device));
    UC_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    UC_HIP(hipEventCreate(&ev[0]));
    UC_HIP(hipEventCreate(&ev[1]));
    GgufFile g;
    gguf_read_header(gguf_path, g);
    Mapped m(gguf_path);
    const std::string arch = g.kv_str.count("general.architecture") ? g.kv_str["general.architecture"] : "t5encoder";
    auto num = [&](const std::string &k, double dflt) { auto it = g.kv_num.find(arch + "." + k); return it == g.kv_num.end() ? dflt : it->second; };
    const GgufTensor *te = find_any(g, {"token_embd.weight", "shared.weight"});
    if (te->ne.size() != 2 || te->ne[0] > 65536 || te->ne[1] > (1u << 24)) fail(UC_ERR_IO, "%s: token embedding is not a [vocab, d_model] matrix", gguf_path.c_str());
    cfg.d_model = (int)te->ne[0];
    cfg.vocab = (int)te->ne[1];
    cfg.n_layers = (int)num("block_count", 24);
    cfg.n_heads = (int)num("attention.head_count", 32);
    cfg.d_kv = (int)num("attention.key_length", cfg.d_model / cfg.n_heads);
    cfg.d_ff = (int)num("feed_forward_length", 16384);
    cfg.rel_buckets = (int)num("attention.relative_buckets_count", 32);
    cfg.rel_max_dist = (int)num("attention.relative_max_distance", 128);
    cfg.eps = (float)num("attention.layer_norm_epsilon", num("attention.layer_norm_rms_epsilon", 1e-6));
    if (cfg.n_layers < 1 || cfg.n_layers > 1024 || cfg.n_heads < 1 || cfg.rel_buckets < 2 || cfg.rel_buckets % 2 || cfg.rel_max_dist < 2 || cfg.d_ff < 64)
        fail(UC_ERR_IO, "%s: implausible encoder geometry (%d layers, %d heads, %d buckets, d_ff %d)", gguf_path.c_str(), cfg.n_layers, cfg.n_heads, cfg.rel_buckets, cfg.d_ff);
    if (const char *e = getenv("UC_T5_EOS_IN_HEAD")) cfg.eos_in_head = atoi(e) != 0;
    if (const char *e = getenv("UC_T5_KEEP_UZOB")) cfg.uzob_to_x = atoi(e) == 0;
    if (cfg.d_kv != 128) fail(UC_ERR_ARGS, "ProstT5 encoder: head size %d not supported (the attention kernel is built for d_kv = 128)", cfg.d_kv);
    if (cfg.d_model % 64 || cfg.d_ff % 64 || (cfg.n_heads * cfg.d_kv) % 64) fail(UC_ERR_ARGS, "ProstT5 encoder: model dimensions must be multiples of 64");

    auto dev_alloc = [&](size_t bytes) { void *p = nullptr; UC_HIP(hipMalloc(&p, std::max<size_t>(bytes, 16))); allocs.push_back(p); return p; };
    auto host_f32 = [&](const GgufTensor *t) {
        std::vector<float> v((size_t)t->n_elems());
        const uint8_t *src = m.p + g.data_offset + t->offset;
        if (t->type == 0) memcpy(v.data(), src, v.size() * 4);
        else for (size_t i = 0; i < v.size(); i++) { uint16_t h; memcpy(&h, src + 2 * i, 2); v[i] = half_to_float(h); }
        return v;
    };
    auto want_vec = [&](const GgufTensor *t, uint64_t n) {
        if (t->n_elems() != n) fail(UC_ERR_IO, "%s: tensor %s has %llu elements, expected %llu", gguf_path.c_str(), t->name.c_str(), (unsigned long long)t->n_elems(), (unsigned long long)n);
        return t;
    };
    auto up_f32 = [&](const std::vector<float> &v) { float *p = (float *)dev_alloc(v.size() * 4); UC_HIP(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return p; };
    // matrix [rows, cols] (ggml ne = [cols, rows]) -> f16 on the device at dst (row-major, cols contiguous)
    auto up_f16_into = [&](const GgufTensor *t, void *dst) {
        const uint8_t *src = m.p + g.data_offset + t->offset;
        const size_t n = (size_t)t->n_elems();
        if (t->type == 1) UC_HIP(hipMemcpy(dst, src, n * 2, hipMemcpyHostToDevice));
        else {
            float *tmp = nullptr;
            UC_HIP(hipMalloc((void **)&tmp, n * 4));
            UC_HIP(hipMemcpy(tmp, src, n * 4, hipMemcpyHostToDevice));
            t5_f32_to_f16(tmp, dst, n, stream);
            UC_HIP(hipStreamSynchronize(stream));
            (void)hipFree(tmp);
        }
    };
    auto want_shape = [&](const GgufTensor *t, uint64_t cols, uint64_t rows) {
        if (t->ne.size() != 2 || t->ne[0] != cols || t->ne[1] != rows)
            fail(UC_ERR_IO, "%s: tensor %s has shape [%llu, %llu], expected [%llu, %llu]", gguf_path.c_str(), t->name.c_str(),
                 (unsigned long long)(t->ne.size() > 0 ? t->ne[0] : 0), (unsigned long long)(t->ne.size() > 1 ? t->ne[1] : 0), (unsigned long long)cols, (unsigned long long)rows);
    };
    const int D = cfg.d_model, HD = cfg.n_heads * cfg.d_kv, F = cfg.d_ff;
    emb = dev_alloc((size_t)cfg.vocab * D * 2);
    up_f16_into(te, emb);
    layers.resize((size_t)cfg.n_layers);
    for (int l = 0; l < cfg.n_layers; l++) {
        const std::string b = "enc.blk." + std::to_string(l) + ".", hf = "encoder.block." + std::to_string(l) + ".layer.";
        const GgufTensor *q = find_any(g, {b + "attn_q.weight", hf + "0.SelfAttention.q.weight"}), *k = find_any(g, {b + "attn_k.weight", hf + "0.SelfAttention.k.weight"}),
                         *v = find_any(g, {b + "attn_v.weight", hf + "0.SelfAttention.v.weight"}), *o = find_any(g, {b + "attn_o.weight", hf + "0.SelfAttention.o.weight"}),
                         *wi = find_any(g, {b + "ffn_up.weight", hf + "1.DenseReluDense.wi.weight"}), *wo = find_any(g, {b + "ffn_down.weight", hf + "1.DenseReluDense.wo.weight"}),
                         *an = find_any(g, {b + "attn_norm.weight", hf + "0.layer_norm.weight"}), *fn = find_any(g, {b + "ffn_norm.weight", hf + "1.layer_norm.weight"});
        want_shape(q, D, HD); want_shape(k, D, HD); want_shape(v, D, HD); want_shape(o, HD, D); want_shape(wi, D, F); want_shape(wo, F, D);
        Layer &L = layers[(size_t)l];
        L.wqkv = dev_alloc((size_t)3 * HD * D * 2);
        up_f16_into(q, L.wqkv);
        up_f16_into(k, (char *)L.wqkv + (size_t)HD * D * 2);
        up_f16_into(v, (char *)L.wqkv + (size_t)2 * HD * D * 2);
        L.wo = dev_alloc((size_t)D * HD * 2); up_f16_into(o, L.wo);
        L.wi = dev_alloc((size_t)F * D * 2); up_f16_into(wi, L.wi);
        L.wo2 = dev_alloc((size_t)D * F * 2); up_f16_into(wo, L.wo2);
        L.attn_norm = up_f32(host_f32(want_vec(an, (uint64_t)D)));
        L.ffn_norm = up_f32(host_f32(want_vec(fn, (uint64_t)D)));
    }
    final_norm = up_f32(host_f32(want_vec(find_any(g, {"enc.output_norm.weight", "encoder.final_layer_norm.weight"}), (uint64_t)D)));
    {   // relative attention bias of block 0 (T5 shares it across blocks): stored [bucket][head] (an Embedding) -> [head][bucket]
        const GgufTensor *rb = find_any(g, {"enc.blk.0.attn_rel_b.weight", "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"});
        const std::vector<float> v = host_f32(rb);
        if (rb->ne.size() != 2 || rb->ne[0] * rb->ne[1] != (uint64_t)cfg.n_heads * cfg.rel_buckets) fail(UC_ERR_IO, "%s: unexpected relative-bias shape", gguf_path.c_str());
        const bool bucket_major = !(rb->ne[0] == (uint64_t)cfg.rel_buckets && rb->ne[1] == (uint64_t)cfg.n_heads && cfg.rel_buckets != cfg.n_heads);
        std::vector<float> hb((size_t)cfg.n_heads * cfg.rel_buckets);
        for (int h = 0; h < cfg.n_heads; h++)
            for (int k = 0; k < cfg.rel_buckets; k++) hb[(size_t)h * cfg.rel_buckets + k] = bucket_major ? v[(size_t)k * cfg.n_heads + h] : v[(size_t)h * cfg.rel_buckets + k];
        rel_bias = up_f32(hb);
    }
    {   // 3Di CNN head: conv1 [hidden, d_model, k], conv2 [n_out, hidden, k] (torch Conv1d / Conv2d (k,1) weight order)
        const GgufTensor *c1 = find_any(g, {"cnn.conv1.weight", "cnn.classifier.0.weight", "classifier.0.weight"}), *b1 = find_any(g, {"cnn.conv1.bias", "cnn.classifier.0.bias", "classifier.0.bias"}),
                         *c2 = find_any(g, {"cnn.conv2.weight", "cnn.classifier.3.weight", "classifier.3.weight"}), *b2 = find_any(g, {"cnn.conv2.bias", "cnn.classifier.3.bias", "classifier.3.bias"});
        if (b1->n_elems() < 1 || b1->n_elems() > 4096 || b2->n_elems() < 1 || b2->n_elems() > 21 || c1->n_elems() < b1->n_elems() * (uint64_t)D)
            fail(UC_ERR_IO, "%s: unexpected CNN head shapes", gguf_path.c_str());
        const std::vector<float> w1 = host_f32(c1), w2 = host_f32(c2);
        cfg.cnn_hidden = (int)b1->n_elems();
        cfg.n_out = (int)b2->n_elems();
        cfg.cnn_kernel = (int)(c1->n_elems() / ((uint64_t)cfg.cnn_hidden * D));
        if (cfg.cnn_kernel < 1 || cfg.cnn_kernel > 31 || !(cfg.cnn_kernel & 1) || (uint64_t)cfg.cnn_hidden * D * cfg.cnn_kernel != c1->n_elems() || (uint64_t)cfg.n_out * cfg.cnn_hidden * cfg.cnn_kernel != c2->n_elems() || cfg.n_out > 21)
            fail(UC_ERR_IO, "%s: unexpected CNN head shapes", gguf_path.c_str());
        const int C1 = cfg.cnn_hidden, KW = cfg.cnn_kernel;
        ldc1 = (KW * C1 + 127) / 128 * 128;
        std::vector<uint16_t> w1r((size_t)ldc1 * D, 0);
        std::vector<float> w1f((size_t)ldc1 * D, 0.f);
        for (int c = 0; c < C1; c++)
            for (int d = 0; d < D; d++)
                for (int k = 0; k < KW; k++) w1f[((size_t)k * C1 + c) * D + d] = w1[((size_t)c * D + d) * KW + k];
        float *tmp = nullptr;
        UC_HIP(hipMalloc((void **)&tmp, w1f.size() * 4));
        UC_HIP(hipMemcpy(tmp, w1f.data(), w1f.size() * 4, hipMemcpyHostToDevice));
        w_conv1 = dev_alloc(w1f.size() * 2);
        t5_f32_to_f16(tmp, w_conv1, w1f.size(), stream);
        UC_HIP(hipStreamSynchronize(stream));
        (void)hipFree(tmp);
        b_conv1 = up_f32(host_f32(b1));
        w_conv2 = up_f32(w2);
        b_conv2 = up_f32(host_f32(b2));
    }
    // tokens: from the file's vocabulary if it carries one, otherwise ProtT5's fixed order (EXT-UNVERIFIED for ProstT5's ids)
    for (int c = 0; c < 256; c++) aa_token[c] = -1;
    if (g.kv_strarr.count("tokenizer.ggml.tokens")) {
        const std::vector<std::string> &tk = g.kv_strarr["tokenizer.ggml.tokens"];
        cfg.prefix_token = -1;
        for (size_t i = 0; i < tk.size(); i++) {
            std::string s = tk[i];
            if (s.size() == 4 && (unsigned char)s[0] == 0xE2 && (unsigned char)s[1] == 0x96 && (unsigned char)s[2] == 0x81) s = s.substr(3);   // U+2581
            if (s.size() == 1 && s[0] >= 'A' && s[0] <= 'Z' && aa_token[(int)s[0]] < 0) aa_token[(int)s[0]] = (int)i;
            if (tk[i] == "<AA2fold>") cfg.prefix_token = (int)i;
            if (tk[i] == "</s>") cfg.eos_token = (int)i;
            if (tk[i] == "<unk>") cfg.unk_token = (int)i;
        }
        if (cfg.prefix_token < 0) fail(UC_ERR_IO, "%s: the vocabulary has no <AA2fold> token", gguf_path.c_str());
    } else {
        const char *order = "ALGVSREDTIPKFQNYMHWCXBOUZ";      // ProtT5 sentencepiece order, ids 3..27
        for (int i = 0; order[i]; i++) aa_token[(int)order[i]] = 3 + i;
        cfg.prefix_token = (int)(g.kv_num.count("prostt5.prefix_token_id") ? g.kv_num["prostt5.prefix_token_id"] : std::min(149, cfg.vocab - 1));
    }
    const int x = aa_token['X'] >= 0 ? aa_token['X'] : cfg.unk_token;
    if (cfg.uzob_to_x) for (char c : {'U', 'Z', 'O', 'B'}) aa_token[(int)c] = x;     // predict_3Di: rare residues are read as X
    for (int c = 'a'; c <= 'z'; c++) aa_token[c] = aa_token[c - 32];
    for (int c = 0; c < 256; c++) if (aa_token[c] < 0) aa_token[c] = x;
    if (g_verbosity >= 3) fprintf(stderr, "ProstT5 encoder: %s: %d layers, d_model %d, %d heads x %d, d_ff %d, vocab %d, CNN %d->%d->%d (k=%d), device %d\n", gguf_path.c_str(), cfg.n_layers,
         cfg.d_model, cfg.n_heads, cfg.d_kv, cfg.d_ff, cfg.vocab, cfg.d_model, cfg.cnn_hidden, cfg.n_out, cfg.cnn_kernel, device);
}

void T5Model::encode_batch(const std::vector<const std::string *> &seqs, std::vector<std::vector<uint8_t>> &out_codes, size_t out_base,
                           std::vector<std::vector<float>> *out_logits) {
    UC_HIP(hipSetDevice(device));
    const int D = cfg.d_model, H = cfg.n_heads, HD = H * cfg.d_kv, F = cfg.d_ff;
    const size_t ns = seqs.size();
    std::vector<int32_t> tok, seq_of, seq_off(ns + 1, 0);
    std::vector<T5AttnTile> tiles;
    int maxL = 1;
    for (size_t s = 0; s < ns; s++) {
        const std::string &a = *seqs[s];
        const int L = (int)a.size() + 2;                     // <AA2fold> residues </s>
        seq_off[s] = (int32_t)tok.size();
        tok.push_back(cfg.prefix_token);
        for (char c : a) tok.push_back(aa_token[(unsigned char)c]);
        tok.push_back(cfg.eos_token);
        for (int i = 0; i < L; i++) seq_of.push_back((int32_t)s);
        for (int q0 = 0; q0 < L; q0 += 128) tiles.push_back({seq_off[s], L, q0});
        maxL = std::max(maxL, L);
    }
    seq_off[ns] = (int32_t)tok.size();
    const int T = (int)tok.size();
    if (!T) return;
    // buffers
    auto grow = [&](void **p, size_t bytes) { if (*p) (void)hipFree(*p); *p = nullptr; UC_HIP(hipMalloc(p, bytes)); };
    if ((size_t)T > cap_tokens) {
        cap_tokens = (size_t)T + (size_t)T / 8 + 256;
        grow((void **)&hidden, cap_tokens * D * 4); grow(&xn, cap_tokens * std::max(D, HD) * 2); grow(&qkv, cap_tokens * 3 * HD * 2); grow(&ao, cap_tokens * HD * 2);
        grow(&ff, cap_tokens * F * 2); grow(&ycnn, cap_tokens * ldc1 * 2); grow((void **)&h1, cap_tokens * cfg.cnn_hidden * 4);
        grow((void **)&logits, cap_tokens * cfg.n_out * 4); grow((void **)&d_tok, cap_tokens * 4); grow((void **)&d_seq_of, cap_tokens * 4);
        grow((void **)&d_codes, cap_tokens);
    }
    if (ns + 1 > cap_seqs) { cap_seqs = ns + ns / 8 + 64; grow((void **)&d_seq_off, cap_seqs * 4); }
    if (tiles.size() > cap_tiles) { cap_tiles = tiles.size() + tiles.size() / 8 + 64; grow((void **)&d_tiles, cap_tiles * sizeof(T5AttnTile)); }
    if (maxL > bias_span) { bias_span = maxL + maxL / 4 + 64; grow((void **)&bias_tab, ((size_t)H * (2 * bias_span - 1) + 8) * 4); }
    {   // bias table per head over key - query in (-span, span): built from the model's bucket table (small: H x (2 span - 1) floats)
        std::vector<float> hb((size_t)H * cfg.rel_buckets);
        UC_HIP(hipMemcpy(hb.data(), rel_bias, hb.size() * 4, hipMemcpyDeviceToHost));
        std::vector<float> tab((size_t)H * (2 * bias_span - 1));
        for (int rel = -(bias_span - 1); rel <= bias_span - 1; rel++) {
            const int b = rel_bucket(rel, cfg.rel_buckets, cfg.rel_max_dist);
            for (int h = 0; h < H; h++) tab[(size_t)h * (2 * bias_span - 1) + (rel + bias_span - 1)] = hb[(size_t)h * cfg.rel_buckets + b];
        }
        UC_HIP(hipMemcpyAsync(bias_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, stream));
        UC_HIP(hipStreamSynchronize(stream));
    }
    UC_HIP(hipMemcpyAsync(d_tok, tok.data(), (size_t)T * 4, hipMemcpyHostToDevice, stream));
    UC_HIP(hipMemcpyAsync(d_seq_of, seq_of.data(), (size_t)T * 4, hipMemcpyHostToDevice, stream));
    UC_HIP(hipMemcpyAsync(d_seq_off, seq_off.data(), (ns + 1) * 4, hipMemcpyHostToDevice, stream));
    UC_HIP(hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(T5AttnTile), hipMemcpyHostToDevice, stream));

    UC_HIP(hipEventRecord(ev[0], stream));
    t5_embed(d_tok, emb, hidden, T, D, cfg.vocab, stream);
    for (int li = 0; li < cfg.n_layers; li++) {
        const Layer &L = layers[(size_t)li];
        t5_rmsnorm(hidden, L.attn_norm, xn, T, D, cfg.eps, stream);
        t5_gemm(0, xn, L.wqkv, qkv, T, 3 * HD, D, stream);
        t5_attention(qkv, d_tiles, (int)tiles.size(), bias_tab, bias_span, H, ao, stream);
        t5_gemm(2, ao, L.wo, hidden, T, D, HD, stream);
        t5_rmsnorm(hidden, L.ffn_norm, xn, T, D, cfg.eps, stream);
        t5_gemm(1, xn, L.wi, ff, T, F, D, stream);
        t5_gemm(2, ff, L.wo2, hidden, T, D, F, stream);
    }
    t5_rmsnorm(hidden, final_norm, xn, T, D, cfg.eps, stream);
    t5_gemm(0, xn, w_conv1, ycnn, T, ldc1, D, stream);
    t5_cnn_head(ycnn, ldc1, d_seq_of, d_seq_off, b_conv1, w_conv2, b_conv2, h1, d_codes, out_logits ? logits : nullptr, T, cfg.cnn_hidden, cfg.cnn_kernel, cfg.n_out, cfg.eos_in_head, stream);
    UC_HIP(hipEventRecord(ev[1], stream));
    std::vector<uint8_t> codes((size_t)T);
    std::vector<float> lg;
    UC_HIP(hipMemcpyAsync(codes.data(), d_codes, (size_t)T, hipMemcpyDeviceToHost, stream));
    if (out_logits) { lg.resize((size_t)T * cfg.n_out); UC_HIP(hipMemcpyAsync(lg.data(), logits, lg.size() * 4, hipMemcpyDeviceToHost, stream)); }
    UC_HIP(hipStreamSynchronize(stream));
    UC_HIP(hipGetLastError());
    float ms = 0;
    UC_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
    stats.total_ms += ms;
    stats.n_seqs += ns;
    stats.n_tokens += (uint64_t)T;
    double attn_flops = 0;
    for (size_t s = 0; s < ns; s++) { const double L = seq_off[s + 1] - seq_off[s]; attn_flops += 4.0 * L * L * HD; }
    stats.flops += (double)cfg.n_layers * (2.0 * T * ((double)3 * HD * D + (double)HD * D + 2.0 * D * F) + attn_flops) + 2.0 * T * (double)ldc1 * D;
    for (size_t s = 0; s < ns; s++) {
        const int b = seq_off[s] + 1, e = seq_off[s + 1] - 1;      // residues only: drop <AA2fold> and </s>
        out_codes[out_base + s].assign(codes.begin() + b, codes.begin() + e);
        if (out_logits) (*out_logits)[out_base + s].assign(lg.begin() + (size_t)b * cfg.n_out, lg.begin() + (size_t)e * cfg.n_out);
    }
}

// batches of similar lengths (sorted, longest first), bounded by tokens: activations are ~(14 d_model + 2 d_ff) bytes per token.  The plan depends on the
// sequence lengths only; a sequence's codes do not depend on the batch it is encoded in (tests/test_configs_gpu.py), so the plan may be worked off by
// any number of model replicas in any order (t5_encode_replicated)
std::vector<std::vector<uint32_t>> t5_plan_batches(const std::vector<std::string> &seqs) {
    const size_t n = seqs.size();
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return seqs[a].size() > seqs[b].size(); });
    size_t tok_budget = 65536;
    if (const char *e = getenv("UC_T5_BATCH_TOKENS")) tok_budget = std::max<size_t>(64, strtoull(e, nullptr, 10));
    std::vector<std::vector<uint32_t>> plan;
    for (size_t i = 0; i < n;) {
        std::vector<uint32_t> ids;
        size_t t = 0;
        while (i < n && (ids.empty() || t + seqs[order[i]].size() + 2 <= tok_budget)) { t += seqs[order[i]].size() + 2; ids.push_back(order[i]); i++; }
        plan.push_back(std::move(ids));
    }
    return plan;
}

void T5Model::encode_ids(const std::vector<std::string> &seqs, const std::vector<uint32_t> &ids, std::vector<std::vector<uint8_t>> &out_codes,
                         std::vector<std::vector<float>> *out_logits) {
    std::vector<const std::string *> batch;
    for (uint32_t i : ids) batch.push_back(&seqs[i]);
    std::vector<std::vector<uint8_t>> tmp_codes(batch.size());
    std::vector<std::vector<float>> tmp_logits;
    if (out_logits) tmp_logits.assign(batch.size(), {});
    encode_batch(batch, tmp_codes, 0, out_logits ? &tmp_logits : nullptr);
    for (size_t k = 0; k < ids.size(); k++) {
        out_codes[ids[k]] = std::move(tmp_codes[k]);
        if (out_logits) (*out_logits)[ids[k]] = std::move(tmp_logits[k]);
    }
}

void T5Model::encode(const std::vector<std::string> &seqs, std::vector<std::vector<uint8_t>> &out_codes, std::vector<std::vector<float>> *out_logits) {
    out_codes.assign(seqs.size(), {});
    if (out_logits) out_logits->assign(seqs.size(), {});
    for (const std::vector<uint32_t> &ids : t5_plan_batches(seqs)) encode_ids(seqs, ids, out_codes, out_logits);
}

// createdb on N GPUs (BASELINE configs[4] on the 8-GPU node; the reference's call is ONE `foldseek createdb`, createdb.rs:157-166): the encoder shards by
// sequence with no exchange at all - "replicas only".  One host thread + one model replica per entry of `devices` (a device may be named more than once:
// several replicas on one GPU, the single-GPU box's test of this path); the length-sorted batch plan is dealt out dynamically, longest batches first (a
// batch's cost grows with the square of its lengths, so a static round-robin would leave the replica that drew the long batches behind); every replica
// writes the codes of its sequences into their slots of `out_codes`, so the output is in input order whatever the dealing was.
void t5_encode_replicated(const std::vector<std::string> &seqs, const std::string &gguf, const std::vector<int> &devices,
                          std::vector<std::vector<uint8_t>> &out_codes, T5Config *cfg_out, T5Stats *stats_out, std::vector<T5Stats> *per_replica) {
    if (devices.empty()) fail(UC_ERR_ARGS, "createdb: no device");
    out_codes.assign(seqs.size(), {});
    const std::vector<std::vector<uint32_t>> plan = t5_plan_batches(seqs);
    const size_t R = devices.size();
    std::vector<T5Stats> rs(R);
    std::vector<T5Config> cfgs(R);
    std::vector<std::string> errs(R);
    std::vector<int> codes_of(R, 0);
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto work = [&](size_t r) {
        try {
            T5Model model;
            model.load(gguf, devices[r]);                      // hipSetDevice is per host thread
            cfgs[r] = model.cfg;
            for (;;) {
                if (failed.load()) break;
                const size_t b = next.fetch_add(1);
                if (b >= plan.size()) break;
                model.encode_ids(seqs, plan[b], out_codes, nullptr);
            }
            rs[r] = model.stats;
        } catch (const Error &f) {
            failed.store(true); errs[r] = f.what(); codes_of[r] = f.code;
        } catch (const std::exception &e) {
            failed.store(true); errs[r] = e.what(); codes_of[r] = UC_ERR_GENERIC;
        }
    };
    if (R == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t r = 0; r < R; r++) th.emplace_back(work, r);
        for (std::thread &t : th) t.join();
    }
    for (size_t r = 0; r < R; r++)
        if (!errs[r].empty()) fail(codes_of[r], "createdb: encoder replica %zu (device %d): %s", r, devices[r], errs[r].c_str());
    T5Stats tot;
    for (const T5Stats &x : rs) {
        tot.n_seqs += x.n_seqs; tot.n_tokens += x.n_tokens; tot.flops += x.flops;
        tot.gemm_ms += x.gemm_ms; tot.attn_ms += x.attn_ms; tot.other_ms += x.other_ms;
        tot.total_ms = std::max(tot.total_ms, x.total_ms);      // the replicas run side by side: the job's GPU time is the slowest replica's
    }
    if (cfg_out) *cfg_out = cfgs[0];
    if (stats_out) *stats_out = tot;
    if (per_replica) *per_replica = rs;
}

// ---------------------------------------------------------------------------------------------- createdb
namespace {
void write_db_files(const std::string &prefix, const std::vector<std::string> &entries, int dbtype) {
    std::ofstream f(prefix, std::ios::binary), ix(prefix + ".index"), dt(prefix + ".dbtype", std::ios::binary);
    if (!f || !ix || !dt) fail(UC_ERR_IO, "cannot write %s", prefix.c_str());
    uint64_t off = 0;
    for (size_t i = 0; i < entries.size(); i++) {
        f.write(entries[i].data(), (std::streamsize)entries[i].size());
        f.write("\n\0", 2);
        ix << i << '\t' << off << '\t' << entries[i].size() + 2 << '\n';
        off += entries[i].size() + 2;
    }
    const uint32_t t = (uint32_t)dbtype;
    dt.write((const char *)&t, 4);
    if (!f || !ix || !dt) fail(UC_ERR_IO, "write error on %s", prefix.c_str());
}
}  // namespace

void t5_createdb(const std::vector<std::string> &fasta_paths, const std::string &out_db, const std::string &model_path, const std::vector<int> &devices, int verbosity,
                 T5Stats *stats_out, std::vector<T5Stats> *per_replica) {
    g_verbosity = verbosity;
    std::vector<std::string> headers, seqs;
    for (const std::string &fp : fasta_paths) {
        std::ifstream in(fp);
        if (!in) fail(UC_ERR_IO, "cannot open %s", fp.c_str());
        std::string line;
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            if (line[0] == '>') { headers.push_back(line.substr(1)); seqs.emplace_back(); }
            else if (!seqs.empty()) { for (char c : line) if (c > ' ') seqs.back().push_back(c >= 'a' && c <= 'z' ? (char)(c - 32) : c); }
        }
    }
    if (seqs.empty()) fail(UC_ERR_ARGS, "createdb: no sequences in the input");
    for (size_t i = 0; i < seqs.size(); i++) if (seqs[i].empty()) fail(UC_ERR_ARGS, "createdb: entry '%s' has no residues", headers[i].c_str());
    std::string gguf = model_path;
    struct stat st;
    if (stat(gguf.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) gguf += "/prostt5-f16.gguf";      // createdb.rs:148
    std::vector<std::vector<uint8_t>> codes;
    Timer tm;
    T5Config cfg;
    T5Stats est;
    t5_encode_replicated(seqs, gguf, devices, codes, &cfg, &est, per_replica);
    logf(3, "ProstT5 encoder: %zu sequences, %llu tokens in %.2f s on %zu replica(s) (weights loaded per replica; %.1f TFLOP/s over all replicas on the GPU timeline)\n",
         seqs.size(), (unsigned long long)est.n_tokens, tm.seconds(), devices.size(), est.total_ms > 0 ? est.flops / (est.total_ms * 1e-3) / 1e12 : 0.0);
    static const char LET[] = "ACDEFGHIKLMNPQRSTVWYX";
    std::vector<std::string> ss(seqs.size());
    for (size_t i = 0; i < seqs.size(); i++) {
        ss[i].resize(codes[i].size());
        for (size_t k = 0; k < codes[i].size(); k++) ss[i][k] = LET[std::min<int>(codes[i][k], 20)];
    }
    write_db_files(out_db, seqs, 0);
    write_db_files(out_db + "_ss", ss, 0);
    {   // which reading of the ProstT5 head produced this 3Di track (both are EXT-UNVERIFIED against Foldseek): databases built under different
        // readings differ in the last ~3 states of every protein and in U/Z/O/B positions - uc_search warns when it is given two that disagree
        std::ofstream sc(out_db + "_ss.source");
        sc << "prostt5_head eos_in_head=" << cfg.eos_in_head << " uzob_to_x=" << cfg.uzob_to_x << "\n";
        if (!sc) fail(UC_ERR_IO, "cannot write %s_ss.source", out_db.c_str());
    }
    write_db_files(out_db + "_h", headers, 12);
    std::ofstream lk(out_db + ".lookup");
    for (size_t i = 0; i < headers.size(); i++) lk << i << '\t' << headers[i].substr(0, headers[i].find_first_of(" \t")) << "\t0\n";
    if (!lk) fail(UC_ERR_IO, "cannot write %s.lookup", out_db.c_str());
    if (stats_out) *stats_out = est;
}

}  // namespace uc
