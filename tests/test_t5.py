"""The ProstT5 AA -> 3Di encoder (BASELINE configs[4]; reference call site createdb.rs:157-166).
PARITY UNPINNED: no ProstT5 weights and no Foldseek here, so the checker is the fp32 PyTorch restatement of the published
architecture (oracle/prostt5_ref.py) on seeded synthetic weights.

Tolerance (f16 operands with fp32 accumulation and an fp32 residual stream against fp32 everywhere): the worst logit error
must stay below 5e-3 of the largest logit of the sequence, and the predicted 3Di state must agree wherever the reference's
top-2 margin exceeds twice that error bound (positions with a thinner margin may legitimately flip: at most 2 % of a sequence, or one residue)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import util

ROOT = util.ROOT
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import prostt5_ref as R  # noqa: E402
import make_t5_golden as G  # noqa: E402

REL_TOL = 5e-3


def _tiny(tmp_path, **kw):
    cfg = R.default_config(**dict(G.CFG, **kw))
    path = str(tmp_path / "t5.gguf")
    R.write_synthetic_gguf(path, cfg, seed=G.SEED)
    return cfg, path


def test_gguf_writer_and_reader_round_trip(tmp_path):
    cfg, path = _tiny(tmp_path)
    kv, w = R.read_gguf(path)
    assert kv["general.architecture"] == "t5encoder" and kv["t5encoder.block_count"] == 2 and kv["t5encoder.embedding_length"] == 128
    assert kv["tokenizer.ggml.tokens"][149] == "<AA2fold>" and kv["tokenizer.ggml.tokens"][3] == "▁A"
    assert w["enc.blk.1.ffn_up.weight"].shape == (512, 128) and w["enc.blk.1.ffn_up.weight"].dtype == np.float16
    assert w["enc.blk.0.attn_rel_b.weight"].shape == (32, 2) and w["cnn.conv1.weight"].shape == (32, 128, 7)
    R.write_synthetic_gguf(str(tmp_path / "again.gguf"), cfg, seed=G.SEED)
    assert open(path, "rb").read() == open(tmp_path / "again.gguf", "rb").read()          # the writer is deterministic


def test_relative_position_buckets_known_answers():
    """T5's bidirectional bucketing, 32 buckets / max distance 128 (Raffel et al. 2020, transformers
    T5Attention._relative_position_bucket): exact below 8, log-spaced up to 128, keys after the query in the upper half"""
    import torch
    rel = torch.tensor([0, -1, 1, -7, 7, -8, 8, -11, -12, -15, -16, -31, -32, -64, -127, -128, -1000, 1000])
    got = R.relative_position_bucket(rel, 32, 128).tolist()
    assert got == [0, 1, 17, 7, 23, 8, 24, 8, 9, 9, 10, 11, 12, 14, 15, 15, 15, 31]


def test_oracle_reproduces_the_committed_fixture(tmp_path):
    cfg, path = _tiny(tmp_path)
    _, w = R.read_gguf(path)
    z = np.load(os.path.join(ROOT, "tests", "golden", "t5_tiny.npz"))
    assert list(z["seqs"]) == G.SEQS
    for i, s in enumerate(G.SEQS):
        lg, codes = R.forward(w, cfg, s)
        assert lg.shape == (len(s), 20) and np.allclose(lg, z["logits%d" % i], rtol=1e-4, atol=1e-4)
        assert np.array_equal(codes, z["codes%d" % i])


def _forward_numpy(w, cfg, seq, eos_in_head=True, uzob_to_x=False):
    """second, torch-free restatement (float64 numpy, explicit loops for the bucket rule and the convolutions)"""
    W = {k: np.asarray(v, np.float64) for k, v in w.items()}
    tok = R.tokenize(seq, cfg, uzob_to_x)
    L, H, dk = len(tok), cfg["n_heads"], cfg["d_kv"]
    h = W["token_embd.weight"][tok]

    def rms(x, g):
        return x / np.sqrt((x * x).mean(-1, keepdims=True) + cfg["eps"]) * g

    def bucket(rel):
        nb = cfg["rel_buckets"] // 2
        r = nb if rel > 0 else 0
        n = abs(rel)
        if n < nb // 2:
            return r + n
        v = nb // 2 + int(np.float32(np.log(np.float32(n) / np.float32(nb // 2))) / np.float32(np.log(cfg["rel_max_dist"] / (nb // 2))) * np.float32(nb - nb // 2))
        return r + min(v, nb - 1)
    bias = np.zeros((H, L, L))
    for i in range(L):
        for j in range(L):
            bias[:, i, j] = W["enc.blk.0.attn_rel_b.weight"][bucket(j - i)]
    for l in range(cfg["n_layers"]):
        b = "enc.blk.%d." % l
        x = rms(h, W[b + "attn_norm.weight"])
        q, k, v = [(x @ W[b + n].T).reshape(L, H, dk).transpose(1, 0, 2) for n in ("attn_q.weight", "attn_k.weight", "attn_v.weight")]
        s_ = np.einsum("hid,hjd->hij", q, k) + bias
        p = np.exp(s_ - s_.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
        h = h + np.einsum("hij,hjd->ihd", p, v).reshape(L, H * dk) @ W[b + "attn_o.weight"].T
        x = rms(h, W[b + "ffn_norm.weight"])
        h = h + np.maximum(x @ W[b + "ffn_up.weight"].T, 0) @ W[b + "ffn_down.weight"].T
    x = rms(h, W["enc.output_norm.weight"])[1:]                      # <AA2fold> off before the head
    if not eos_in_head:
        x[-1] = 0                                                    # predict_3Di reading: </s> masked to zero, its position stays
    KW = cfg["cnn_kernel"]

    def conv(x, w_, b_):                                             # x [n, cin], w_ [cout, cin, k]: cross-correlation, zero padding
        n = len(x)
        y = np.tile(b_, (n, 1))
        for t in range(n):
            for k in range(KW):
                u = t + k - KW // 2
                if 0 <= u < n:
                    y[t] += w_[:, :, k] @ x[u]
        return y
    y = conv(np.maximum(conv(x, W["cnn.conv1.weight"], W["cnn.conv1.bias"]), 0), W["cnn.conv2.weight"], W["cnn.conv2.bias"])
    return y[:-1]                                                    # </s> off after it


def test_oracle_against_a_second_independent_restatement(tmp_path):
    """the torch restatement (the checker of the GPU tests) agrees with a torch-free float64 numpy one: bucket rule, shared
    bias of block 0, un-scaled attention, pre-norm residual blocks, and BOTH head conventions (prefix off before the convolutions,
    </s> dropped after them; default: </s>'s hidden state feeds the CNN, B/O/U/Z keep their ids; the predict_3Di reading: </s>
    zeroed before the CNN, U/Z/O/B -> X)"""
    cfg, path = _tiny(tmp_path)
    _, w = R.read_gguf(path)
    for seq in ("M", "MKTAYIAKQRQISFVKSH", "ACDEFGHIKLMNPQRSTVWYXBZ" * 7):
        for conv in (dict(eos_in_head=True, uzob_to_x=False), dict(eos_in_head=False, uzob_to_x=True)):
            lg, codes = R.forward(w, cfg, seq, **conv)
            ref = _forward_numpy(w, cfg, seq, **conv)
            assert ref.shape == lg.shape and np.abs(ref - lg).max() <= 2e-4 * max(np.abs(ref).max(), 1.0), (seq, conv)
            assert (ref.argmax(1) == codes).mean() > 0.99
    a, _ = R.forward(w, cfg, "MKTAYIAKQRUZOBQISFVKSH")
    b, _ = R.forward(w, cfg, "MKTAYIAKQRUZOBQISFVKSH", eos_in_head=False, uzob_to_x=True)
    assert np.abs(a - b).max() > 1e-3                                  # the two readings do differ


def test_encoder_fails_loudly_without_a_gpu(tmp_path):
    import unicore_amd as U
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _, path = _tiny(tmp_path)
    with pytest.raises(U.UcError) as ei:
        U.T5Encoder(path)
    assert ei.value.code == U.UC_ERR_DEVICE and "no CPU fallback" in str(ei.value)
    with pytest.raises(U.UcError) as ei:
        U.T5Encoder(str(tmp_path / "missing.gguf"))
    shim = os.path.join(ROOT, "bin", "foldseek")
    r = subprocess.run([shim, "createdb", "in.fasta", "out_db"], capture_output=True, text=True)
    assert r.returncode == 2 and "--prostt5-model" in r.stderr


# ---------------------------------------------------------------------------------------------- GPU
def _check(codes, logits, ref_logits, ref_codes, what):
    err = float(np.abs(logits - ref_logits).max())
    scale = float(np.abs(ref_logits).max())
    assert err <= REL_TOL * scale, (what, err, scale)
    top2 = np.sort(ref_logits, axis=1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 2 * REL_TOL * scale
    assert np.array_equal(codes[safe], ref_codes[safe]), what
    assert (codes != ref_codes).sum() <= max(1, 0.02 * len(ref_codes)), what       # thin-margin positions may flip: at most 2 % (or one residue)


@pytest.mark.gpu
def test_hip_encoder_against_the_fixture_and_the_fp32_restatement(tmp_path):
    import unicore_amd as U
    cfg, path = _tiny(tmp_path)
    z = np.load(os.path.join(ROOT, "tests", "golden", "t5_tiny.npz"))
    enc = U.T5Encoder(path)
    codes, logits = enc.encode(G.SEQS, logits=True)
    for i, s in enumerate(G.SEQS):
        assert logits[i].shape == (len(s), 20)
        _check(codes[i], logits[i], z["logits%d" % i], z["codes%d" % i], "fixture %d" % i)
    # batching must not matter: one sequence at a time gives the same bytes as the batch
    for i in (0, 2, 3):
        c1, l1 = enc.encode([G.SEQS[i]], logits=True)
        assert np.array_equal(c1[0], codes[i]) and np.array_equal(l1[0], logits[i])
    st = enc.stats()
    assert st["n_tokens"] > 0 and st["flops"] > 0 and st["gpu_ms"] > 0
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [dict(d_model=256, n_heads=4, d_ff=1024, n_layers=3), dict(d_model=1024, n_heads=32, d_ff=16384, n_layers=2)])
def test_hip_encoder_wider_models_and_ragged_batches(geom, tmp_path):
    """ProtT5-XL width (1024 / 32 heads / 16384) with few blocks, lengths around the tile edges (63, 64, 65, 127, 129), a
    1-residue sequence, non-standard residues, and small token batches (UC_T5_BATCH_TOKENS) against one big batch"""
    import unicore_amd as U
    cfg = R.default_config(d_kv=128, **geom)
    path = str(tmp_path / "m.gguf")
    R.write_synthetic_gguf(path, cfg, seed=0x5EED0006, with_vocab=geom["n_layers"] == 3)      # with and without a vocabulary in the file
    _, w = R.read_gguf(path)
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), L)) for L in (1, 2, 63, 64, 65, 127, 129, 300)] + ["MKXXBZOUACDEFGHIKLMNPQRSTVWYmkt"]
    enc = U.T5Encoder(path)
    codes, logits = enc.encode(seqs, logits=True)
    for s, c, lg in zip(seqs, codes, logits):
        rl, rc = R.forward(w, cfg, s)
        _check(c, lg, rl, rc, (geom, len(s)))
    os.environ["UC_T5_BATCH_TOKENS"] = "200"
    try:
        c2, l2 = enc.encode(seqs, logits=True)
    finally:
        del os.environ["UC_T5_BATCH_TOKENS"]
    for a, b, la, lb in zip(codes, c2, logits, l2):
        assert np.array_equal(a, b) and np.array_equal(la, lb)
    enc.close()


@pytest.mark.gpu
def test_hip_encoder_large_batches_use_the_256_tile_gemm_with_identical_results(tmp_path):
    """batches of >= 2048 tokens run the linear layers on the 256 x 256 tile kernel, smaller ones on the 128 x 128 one: the K order
    per output element is the same, so states AND logits must be bit-identical; a few sequences also go against the fp32
    restatement"""
    import unicore_amd as U
    cfg = R.default_config(d_kv=128, d_model=1024, n_heads=32, d_ff=16384, n_layers=1)
    path = str(tmp_path / "m.gguf")
    R.write_synthetic_gguf(path, cfg, seed=0x5EED0008)
    _, w = R.read_gguf(path)
    rng = np.random.default_rng(9)
    seqs = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), L)) for L in (400, 391, 380, 333, 300, 290, 257, 255, 129, 64, 17)]   # 2816 residues
    enc = U.T5Encoder(path)
    codes, logits = enc.encode(seqs, logits=True)                     # one batch of 2838 tokens -> 256 tile
    os.environ["UC_T5_BATCH_TOKENS"] = "900"                          # batches below 2048 tokens -> 128 tile
    try:
        c2, l2 = enc.encode(seqs, logits=True)
    finally:
        del os.environ["UC_T5_BATCH_TOKENS"]
    for a, b, la, lb in zip(codes, c2, logits, l2):
        assert np.array_equal(a, b) and np.array_equal(la, lb)
    for i in (0, 7, 10):
        rl, rc = R.forward(w, cfg, seqs[i])
        _check(codes[i], logits[i], rl, rc, ("256 tile", len(seqs[i])))
    enc.close()


@pytest.fixture(scope="module")
def full_model():
    """the full-size synthetic model file (ProtT5-XL geometry, 24 blocks; shared with bench.py --config c5 through /tmp/uc_bench)"""
    import make_t5_full_depth as F
    return F, F.ensure_gguf()


@pytest.mark.gpu
def test_hip_encoder_full_depth_24_blocks(full_model):
    """BASELINE configs[4]'s model as bench.py times it — ProtT5-XL geometry AND all 24 blocks — against (a) the committed fp32
    fixture tests/golden/t5_full_depth.npz (generated in the build container, no torch needed for it) and (b) the fp32 PyTorch
    restatement run on this host for lengths 1 ... 1000 (tile edges 63 / 64 / 65 / 129, one batch, ragged).  Same tolerance as the
    shallow tests: worst logit error <= 5e-3 of the sequence's largest logit, states equal wherever the margin allows."""
    import unicore_amd as U
    F, path = full_model
    z = np.load(os.path.join(ROOT, "tests", "golden", "t5_full_depth.npz"))
    assert list(z["seqs"]) == F.SEQS
    enc = U.T5Encoder(path)
    codes, logits = enc.encode(F.SEQS, logits=True)
    for i, s in enumerate(F.SEQS):
        assert logits[i].shape == (len(s), 20)
        _check(codes[i], logits[i], z["logits%d" % i], z["codes%d" % i], "full-depth fixture %d" % i)
    cfg = R.default_config()
    _, w = R.read_gguf(path)
    W = R.prepare(w)
    rng = np.random.default_rng(24)
    seqs = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), L)) for L in (1, 2, 63, 64, 65, 129, 257, 400, 700, 1000)]
    codes, logits = enc.encode(seqs, logits=True)
    worst = 0.0
    for s, c, lg in zip(seqs, codes, logits):
        rl, rc = R.forward(W, cfg, s)
        _check(c, lg, rl, rc, ("24 blocks", len(s)))
        worst = max(worst, float(np.abs(lg - rl).max() / np.abs(rl).max()))
    print("full depth: worst relative logit error %.2e" % worst)
    # the head-convention switches reach the product the same way they reach the oracle (EXT-UNVERIFIED table): the OTHER
    # reading (ProstT5's predict_3Di script) against its own committed fixture and against the restatement
    enc.close()
    os.environ["UC_T5_EOS_IN_HEAD"] = "0"
    os.environ["UC_T5_KEEP_UZOB"] = "0"
    try:
        enc2 = U.T5Encoder(path)
        codes, logits = enc2.encode(F.SEQS, logits=True)
        for i in range(len(F.SEQS)):
            _check(codes[i], logits[i], z["logits%d_p3d" % i], z["codes%d_p3d" % i], "full-depth fixture %d, predict_3Di reading" % i)
        s = "MKTAYIAKQRUZOBQISFVKSH"
        c2, l2 = enc2.encode([s], logits=True)
        rl, rc = R.forward(W, cfg, s, eos_in_head=False, uzob_to_x=True)
        _check(c2[0], l2[0], rl, rc, "predict_3Di head convention")
        rl1, _ = R.forward(W, cfg, s)
        assert np.abs(rl - rl1).max() > 1e-3                     # the two conventions do differ (last residues, U/Z/O/B)
        enc2.close()
    finally:
        del os.environ["UC_T5_EOS_IN_HEAD"], os.environ["UC_T5_KEEP_UZOB"]


@pytest.mark.gpu
def test_gguf_loader_rejects_malformed_files(tmp_path):
    """a truncated or foreign prostt5-f16.gguf must fail with UC_ERR_IO, not crash or read out of bounds"""
    import struct
    import unicore_amd as U
    cfg, path = _tiny(tmp_path)
    good = open(path, "rb").read()
    kv, w = R.read_gguf(path)

    def expect_io(data, what):
        bad = str(tmp_path / "bad.gguf")
        open(bad, "wb").write(data)
        with pytest.raises(U.UcError) as ei:
            U.T5Encoder(bad)
        assert ei.value.code == U.UC_ERR_IO, (what, str(ei.value))
    expect_io(good[: len(good) // 2], "truncated")
    expect_io(b"GGML" + good[4:], "foreign magic")
    # alignment 0 / not a power of two
    for al in (0, 48):
        p2 = str(tmp_path / "al.gguf")
        tens = [(k, np.array(v)) for k, v in w.items()]
        R.write_gguf(p2, dict(kv, **{"general.alignment": al}), tens)
        expect_io(open(p2, "rb").read(), "alignment %d" % al)
    # a norm vector of the wrong length, a 1-d token embedding
    for name, arr, what in (("enc.blk.0.attn_norm.weight", np.ones(7, np.float32), "short norm"),
                            ("token_embd.weight", np.ones(150 * 128, np.float16), "1-d embedding"),
                            ("cnn.conv1.bias", np.zeros(0, np.float32), "empty bias")):
        p2 = str(tmp_path / "shape.gguf")
        tens = [(k, arr if k == name else np.array(v)) for k, v in w.items()]
        if arr.size == 0:
            continue                                               # the writer cannot express a zero dim; covered by the header patch below
        R.write_gguf(p2, kv, tens)
        expect_io(open(p2, "rb").read(), what)
    # a zero dimension patched into the first tensor's header
    name = b"token_embd.weight"
    i = good.index(name) + len(name)
    nd = struct.unpack_from("<I", good, i)[0]
    assert nd == 2
    expect_io(good[: i + 4] + struct.pack("<Q", 0) + good[i + 12:], "zero dim")


@pytest.mark.gpu
def test_createdb_writes_the_database_the_cluster_path_reads(tmp_path):
    """`foldseek createdb <fasta> <db> --prostt5-model <dir>` (createdb.rs:157-166) through the shim, then `foldseek
    cluster` + `createtsv` on the result: the whole chain of `unicore createdb` -> `unicore cluster` without Foldseek"""
    import unicore_amd as U
    cfg, path = _tiny(tmp_path)
    mdir = tmp_path / "weights"
    mdir.mkdir()
    os.rename(path, mdir / "prostt5-f16.gguf")                    # createdb.rs:148: <model>/prostt5-f16.gguf
    rng = np.random.default_rng(9)
    fams = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), int(rng.integers(60, 200)))) for _ in range(6)]
    recs = []
    for f, a in enumerate(fams):
        for m in range(4):
            s = list(a)
            for p in rng.choice(len(s), len(s) // 12, replace=False):
                s[p] = rng.choice(list("ACDEFGHIKLMNPQRSTVWY"))
            recs.append(("unicore_%010x fam%d member%d" % (f * 16 + m + 1, f, m), "".join(s)))
    fa = tmp_path / "combined_aa.fasta"
    fa.write_text("".join(">%s\n%s\n" % (h, "\n".join(s[i:i + 60] for i in range(0, len(s), 60))) for h, s in recs))
    shim = os.path.join(ROOT, "bin", "foldseek")
    db = str(tmp_path / "proteome_db")
    r = subprocess.run([shim, "createdb", str(fa), db, "--prostt5-model", str(mdir), "--threads", "4", "--gpu", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for sfx in ("", "_ss", "_h", ".index", "_ss.index", "_h.index", ".dbtype", "_ss.dbtype", "_h.dbtype", ".lookup"):
        assert os.path.exists(db + sfx), sfx
    aa = open(db, "rb").read().split(b"\n\0")[:-1]
    ss = open(db + "_ss", "rb").read().split(b"\n\0")[:-1]
    assert [a.decode() for a in aa] == [s for _, s in recs]
    assert all(len(a) == len(b) and set(b.decode()) <= set("ACDEFGHIKLMNPQRSTVWY") for a, b in zip(aa, ss))
    # the 3Di track is what the library's own encoder predicts
    _, w = R.read_gguf(str(mdir / "prostt5-f16.gguf"))
    enc = U.T5Encoder(str(mdir))
    codes = enc.encode([s for _, s in recs])
    assert ["".join("ACDEFGHIKLMNPQRSTVWY"[c] for c in cs) for cs in codes] == [b.decode() for b in ss]
    enc.close()
    # ... and the cluster path runs on it (members of one family differ in ~8 % of their residues)
    out = str(tmp_path / "clust")
    subprocess.run([shim, "cluster", "--threads", "4", "-v", "1", db, out + "_cluster", str(tmp_path / "tmp"), "-c", "0.8"], check=True)
    subprocess.run([shim, "createtsv", "--threads", "4", "-v", "1", db, db, out + "_cluster", out + ".tsv"], check=True)
    names = [h.split()[0] for h, _ in recs]
    rows = util.tsv_invariants(out + ".tsv", names)
    assert len(rows) == len(recs)
    # the reading of the ProstT5 head that predicted the track travels with the database ...
    assert open(db + "_ss.source").read() == "prostt5_head eos_in_head=1 uzob_to_x=0\n"
    # ... and a search across two databases built under different readings says so (and runs)
    db2 = str(tmp_path / "proteome_db_p3d")
    r = subprocess.run([shim, "createdb", str(fa), db2, "--prostt5-model", str(mdir), "--threads", "4"], capture_output=True, text=True,
                       env=dict(os.environ, UC_T5_EOS_IN_HEAD="0", UC_T5_KEEP_UZOB="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(db2 + "_ss.source").read() == "prostt5_head eos_in_head=0 uzob_to_x=1\n"
    for q, want in ((db, False), (db2, True)):
        r = subprocess.run([shim, "search", "--threads", "4", "-v", "2", q, db, str(tmp_path / "aln"), str(tmp_path / "tmp")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("different ProstT5 head conventions" in r.stdout + r.stderr) == want, (r.stdout + r.stderr)[-600:]


def _ndev():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def _createdb_case(tmp_path, n_seqs=40, seed=31):
    cfg, path = _tiny(tmp_path)
    rng = np.random.default_rng(seed)
    recs = [("unicore_%010x entry %d" % (k + 1, k), "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), int(rng.integers(30, 260))))) for k in range(n_seqs)]
    fa = tmp_path / "in.fasta"
    fa.write_text("".join(">%s\n%s\n" % r for r in recs))
    return path, str(fa), recs


DB_FILES = ("", "_ss", "_h", ".index", "_ss.index", "_h.index", ".dbtype", "_ss.dbtype", "_h.dbtype", ".lookup", "_ss.source")


@pytest.mark.gpu
def test_createdb_on_n_encoder_replicas_writes_the_same_database(tmp_path, monkeypatch):
    """createdb on N GPUs (BASELINE configs[4] on the 8-GPU node; the reference's ONE `foldseek createdb` call, createdb.rs:157-166): one encoder
    replica per GPU, the sequences sharded over them by a dynamic deal of the length-sorted batch plan, no collective.  On the single-GPU box the
    replicas share the device (UC_VIRTUAL_GPUS=1, as for uc_cluster's virtual ranks): every file of the database is byte-identical to the one
    replica's, whatever N and whatever the dealing was; without the switch, asking for more GPUs than are visible is an error, not a silent fallback."""
    import unicore_amd as U
    model, fa, recs = _createdb_case(tmp_path)
    monkeypatch.setenv("UC_T5_BATCH_TOKENS", "700")          # ~12 batches: something to deal out
    ref = str(tmp_path / "db1")
    st1 = U.createdb(fa, ref, model, num_gpus=1)
    assert st1["n_replicas"] == 1 and st1["n_seqs"] == len(recs) and st1["tokens_min_replica"] == st1["tokens_max_replica"] == st1["n_tokens"]
    want = {s: open(ref + s, "rb").read() for s in DB_FILES}
    assert want["_ss"] and len(want["_ss"]) == len(want[""])
    nd = _ndev()
    if nd < 4:
        with pytest.raises(U.UcError) as ei:
            U.createdb(fa, str(tmp_path / "dbx"), model, num_gpus=4)
        assert ei.value.code == 4 and "GPUs requested" in str(ei.value)
    monkeypatch.setenv("UC_VIRTUAL_GPUS", "1")
    for n in (2, 3, 5):
        out = str(tmp_path / ("db%d" % n))
        st = U.createdb(fa, out, model, num_gpus=n)
        assert st["n_replicas"] == n and st["n_seqs"] == len(recs) and st["n_tokens"] == st1["n_tokens"] and abs(st["flops"] - st1["flops"]) <= 1e-9 * st1["flops"]
        assert st["tokens_max_replica"] <= st["n_tokens"] and st["gpu_ms"] <= st["gpu_ms_sum"] + 1e-9
        for s in DB_FILES:
            assert open(out + s, "rb").read() == want[s], (n, s)
    # the shim's spelling: `--gpus N` next to the reference's own arguments
    shim = os.path.join(ROOT, "bin", "foldseek")
    out = str(tmp_path / "db_shim")
    r = subprocess.run([shim, "createdb", fa, out, "--prostt5-model", model, "--threads", "4", "--gpu", "1", "--gpus", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "3 replica(s)" in r.stdout + r.stderr
    for s in DB_FILES:
        assert open(out + s, "rb").read() == want[s], ("shim", s)
    # a replica that fails takes the call down with its message (here: every replica - the model file is gone)
    with pytest.raises(U.UcError):
        U.createdb(fa, str(tmp_path / "dby"), str(tmp_path / "no_such.gguf"), num_gpus=2)


@pytest.mark.gpu
@pytest.mark.skipif(_ndev() < 2, reason="needs >= 2 visible GPUs (one encoder replica per GPU); replicas sharing one GPU are covered by the test above")
def test_createdb_on_all_visible_gpus(tmp_path, monkeypatch):
    """the same on REAL devices: num_gpus = 0 (all visible) == one replica, file for file"""
    import unicore_amd as U
    model, fa, recs = _createdb_case(tmp_path, n_seqs=120)
    monkeypatch.setenv("UC_T5_BATCH_TOKENS", "700")
    ref, out = str(tmp_path / "db1"), str(tmp_path / "dbN")
    U.createdb(fa, ref, model, num_gpus=1)
    st = U.createdb(fa, out, model, num_gpus=0)
    assert st["n_replicas"] == _ndev()
    for s in DB_FILES:
        assert open(out + s, "rb").read() == open(ref + s, "rb").read(), s
