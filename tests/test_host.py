"""CPU-side tests of the product's host logic and C-ABI surface (no compute calls — there is no GPU here)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import util
import unicore_amd as U
from oracle import oracle_py as O

ROOT = util.ROOT


def test_library_loads_and_exports_every_declared_symbol():
    L = U.lib()
    hdr = open(os.path.join(ROOT, "include", "unicore_cluster.h")).read()
    declared = set(re.findall(r"\b(uc_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(U.SYMBOLS), declared ^ set(U.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert "gfx950" in U.version()
    assert os.path.exists(os.path.join(ROOT, "unicore_amd", "libunicore_cluster.so"))


def test_no_gpu_means_loud_failure_not_fallback():
    """the product path must fail loudly without the device: error class 4, never a CPU fallback"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(U.UcError) as ei:
        U.Engine("-c 0.8")
    assert ei.value.code == U.UC_ERR_DEVICE and "no CPU fallback" in str(ei.value)
    with pytest.raises(U.UcError) as ei:
        U.cluster("/nonexistent/db", "/tmp/x_cluster", "/tmp/x_tmp")
    assert ei.value.code in (U.UC_ERR_DEVICE, U.UC_ERR_IO)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under unicore_amd/ or include/ may reference it"""
    bad = []
    for base in ("unicore_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"uc_oracle|oracle_py|liboracle|from oracle|import oracle|uco_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = subprocess.run(["ldd", U.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


@pytest.mark.parametrize("opts,ok", [
    ("-c 0.8", True), ("", True), ("-c 0.8 --min-seq-id 0.3 -s 7.5 -e 0.001 --cov-mode 1 --max-seqs 100", True),
    ("--single-step-clustering --cluster-mode 0 --alignment-type 2 --threads 4 -v 2", True),
    ("--single-step-clustering 1 -c 0.5", True), ("--k-score 25 --min-ungapped-score 20 --gap-open 11 --gap-extend 1", True),
    ("--bogus 1", False), ("-c", False), ("-c abc", False), ("-c 1.5", False), ("--cov-mode 5", False),
    ("--cluster-mode 1", False), ("--alignment-type 1", False), ("--max-seqs 0", False), ("stray", False),
])
def test_foldseek_style_option_parser(opts, ok):
    rc = U.check_options(opts)
    assert (rc == 0) == ok
    if not ok:
        assert rc == U.UC_ERR_ARGS and len(U.lib().uc_last_error()) > 0


def test_product_setcover_equals_oracle_on_random_graphs():
    rng = np.random.default_rng(11)
    for _ in range(80):
        n = int(rng.integers(1, 300))
        m = int(rng.integers(0, 4 * n))
        e = rng.integers(0, n, (m, 2)).astype(np.uint32)
        a, b = U.setcover(n, e), O.setcover(n, e)
        assert np.array_equal(a, b)
        assert (a[a] == a).all()
    # clique-ish + hub structure at a larger size
    n = 6000
    fam = rng.integers(0, 150, n)
    e = np.array([(i, j) for i in range(0, n, 7) for j in np.nonzero(fam == fam[i])[0][:20]], np.uint32)
    assert np.array_equal(U.setcover(n, e), O.setcover(n, e))
    with pytest.raises(U.UcError):
        U.setcover(3, [(0, 5)])


def _random_lists(rng, n, m, max_t):
    cnt = rng.integers(0, m + 1, n).astype(np.uint32)
    hits = np.zeros(int(cnt.sum()), U.HIT_DTYPE)
    k = 0
    for q in range(n):
        t = rng.choice(max_t, cnt[q], replace=False)
        sc = rng.integers(15, 256, cnt[q])
        order = np.lexsort((t, -sc))
        hits["target"][k:k + cnt[q]] = t[order]
        hits["score"][k:k + cnt[q]] = sc[order]
        hits["diag"][k:k + cnt[q]] = rng.integers(-50, 50, cnt[q])
        k += cnt[q]
    return cnt, hits


def test_hits_merge_is_independent_of_the_sharding():
    """split every query's list by target range into 1/2/4/8 'shards', truncate each to M, merge: the
    result equals the top-M of the full list (per-shard truncation is lossless) — SURVEY.md 8(e)"""
    rng = np.random.default_rng(2)
    n, M, max_t = 200, 12, 500
    cnt, hits = _random_lists(rng, n, 40, max_t)
    off = np.concatenate([[0], np.cumsum(cnt.astype(np.int64))]).astype(np.int64)
    expect_c, expect_h = U.hits_merge(n, M, [(cnt, hits)])
    assert (expect_c == np.minimum(cnt, M)).all()
    for shards in (2, 4, 8):
        bounds = np.linspace(0, max_t, shards + 1).astype(int)
        parts = []
        for s in range(shards):
            pc, ph = np.zeros(n, np.uint32), []
            for q in range(n):
                h = hits[off[q]:off[q + 1]]
                h = h[(h["target"] >= bounds[s]) & (h["target"] < bounds[s + 1])][:M]
                pc[q] = len(h)
                ph.append(h)
            parts.append((pc, np.concatenate(ph) if ph else np.zeros(0, U.HIT_DTYPE)))
        c, h = U.hits_merge(n, M, parts)
        assert np.array_equal(c, expect_c) and np.array_equal(h, expect_h)
    # duplicated target across shards is a caller bug and must be rejected, not merged silently
    with pytest.raises(U.UcError):
        U.hits_merge(1, 5, [(np.array([1], np.uint32), hits[:1]), (np.array([1], np.uint32), hits[:1])])


def test_cluster_db_createtsv_rmdb_roundtrip(tmp_path):
    s3, sa = util.family_db(4, n_fam=6, members=4, with_x=False)
    names = util.write_db(str(tmp_path / "db"), s3, sa)
    n = len(s3)
    rng = np.random.default_rng(0)
    reps = np.sort(rng.choice(n, 7, replace=False))
    assign = reps[rng.integers(0, len(reps), n)].astype(np.uint32)
    assign[reps] = reps
    cdb = str(tmp_path / "out" / "clust_cluster")
    os.makedirs(os.path.dirname(cdb))
    assert U.lib().uc_write_cluster_db(cdb.encode(), n, assign.ctypes.data) == 0
    for sfx in ("", ".index", ".dbtype"):
        assert os.path.exists(cdb + sfx)
    assert int.from_bytes(open(cdb + ".dbtype", "rb").read(), "little") == 6
    # cluster DB entry layout: member keys one per line, representative first, NUL-terminated
    first = open(cdb, "rb").read().split(b"\0")[0].decode().split()
    assert int(first[0]) == reps[0] and sorted(map(int, first[1:])) == list(map(int, first[1:]))
    tsv = str(tmp_path / "out" / "clust.tsv")
    U.createtsv(str(tmp_path / "db"), cdb, tsv)
    rows = util.tsv_invariants(tsv, names)
    assert [r[0] for r in rows] == [names[assign[i]] for i in np.lexsort((np.arange(n), np.arange(n) != assign, assign))]
    # the oracle's writer produces the same bytes
    odb = O.OracleDb(str(tmp_path / "db"))
    O.write_tsv(str(tmp_path / "ref.tsv"), odb, assign)
    assert open(tsv, "rb").read() == open(str(tmp_path / "ref.tsv"), "rb").read()
    U.rmdb(cdb)
    assert not any(os.path.exists(cdb + sfx) for sfx in ("", ".index", ".dbtype"))
    bad = assign.copy(); bad[reps[0]] = reps[1]          # a cluster whose representative is not its own member
    assert U.lib().uc_write_cluster_db(cdb.encode(), n, bad.ctypes.data) != 0


def test_cli_surface_and_checkpoint_contract(tmp_path):
    """`unicore cluster` mirrors src/util/arg_parser.rs:225-246 + src/modules/cluster.rs: no arguments -> the help text
    (clap's arg_required_else_help, arg_parser.rs:8,226), clap usage errors exit 2, the checkpoint is written as "0"
    before the engine runs, engine failure -> exit 1 with 'Error: ...'"""
    exe = os.path.join(ROOT, "bin", "unicore")
    for argv in ([exe], [exe, "cluster"]):
        r = subprocess.run(argv, capture_output=True, text=True)
        assert r.returncode == 2 and "Usage: unicore cluster [OPTIONS] <INPUT> <OUTPUT> <TMP>" in r.stderr and "--keep-cluster-db" in r.stderr
    r = subprocess.run([exe, "cluster", "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage: unicore cluster" in r.stdout
    r = subprocess.run([exe, "cluster", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 2 and "required arguments were not provided" in r.stderr
    r = subprocess.run([exe, "cluster", "a", "b", "c", "--nope"], capture_output=True, text=True)
    assert r.returncode == 2 and "unexpected argument '--nope'" in r.stderr
    r = subprocess.run([exe, "version"], capture_output=True, text=True)
    assert r.returncode == 0 and "unicore-cluster" in r.stdout
    out = tmp_path / "res" / "clust"
    r = subprocess.run([exe, "cluster", str(tmp_path / "missing_db"), str(out), str(tmp_path / "tmp"), "-c", "-c 0.8", "-v", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "\nError: " in "\n" + r.stderr
    assert open(tmp_path / "res" / "cluster.chk").read() == "0"      # started, never finished
    r = subprocess.run([exe, "cluster", "db", str(out), "tmp", "-c", "--frobnicate 3"], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown or unsupported cluster option" in r.stderr
    # `unicore search` (arg_parser.rs:247-270, search.rs): same contract, its own checkpoint file
    r = subprocess.run([exe, "search", "a", "b", "c"], capture_output=True, text=True)
    assert r.returncode == 2
    sout = tmp_path / "sres" / "hits"
    r = subprocess.run([exe, "search", str(tmp_path / "missing_q"), str(tmp_path / "missing_t"), str(sout), str(tmp_path / "tmp"), "-s", "-c 0.8", "-v", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "\nError: " in "\n" + r.stderr
    assert open(tmp_path / "sres" / "search.chk").read() == "0"
    shim = os.path.join(ROOT, "bin", "foldseek")
    assert subprocess.run([shim, "version"], capture_output=True).returncode == 0      # config.rs:49-66 handshake
    assert subprocess.run([shim, "search", "a", "b"], capture_output=True).returncode != 0
    assert subprocess.run([shim, "convertalis", "a"], capture_output=True).returncode != 0
    assert subprocess.run([shim, "easy-search"], capture_output=True).returncode != 0
    r = subprocess.run([shim, "rmdb", str(tmp_path / "nothing_cluster"), "-v", "2"], capture_output=True)
    assert r.returncode == 0


def test_shim_accepts_flags_anywhere(tmp_path):
    """`foldseek cluster` flags may stand before, between or after the positionals; whether a flag takes
    a value comes from the engine's flag table (uc_option_arity), not from its position.  Without a GPU every well-formed
    command line reaches the engine and fails there (status 3: the DB does not exist / 4: no device), a malformed one
    fails in the parser (status 2)."""
    shim = os.path.join(ROOT, "bin", "foldseek")
    db, out, tmp = str(tmp_path / "nodb"), str(tmp_path / "o_cluster"), str(tmp_path / "t")
    L = U.lib()
    assert L.uc_option_arity(b"-c") == 1 and L.uc_option_arity(b"--single-step-clustering") == 2 and L.uc_option_arity(b"--nope") == -1
    for argv in (["cluster", "--threads", "2", "-v", "1", db, out, tmp, "-c", "0.8"],                 # cluster.rs:45-49 order
                 ["cluster", "-c", "0.8", db, out, tmp],                                             # valued flag first
                 ["cluster", db, "-c", "0.8", "--threads", "2", out, "--min-seq-id", "0.3", tmp, "-v", "1"],
                 ["cluster", "--single-step-clustering", db, out, tmp, "-e", "1e-3"],                # switch before a positional
                 ["cluster", "--single-step-clustering", "1", db, out, tmp]):
        r = subprocess.run([shim] + argv, capture_output=True, text=True)
        assert r.returncode in (3, 4) and ("cannot" in r.stderr or "HIP" in r.stderr or "device" in r.stderr), (argv, r.returncode, r.stderr)
    for argv in (["cluster", db, out], ["cluster", db, out, tmp, "extra"], ["cluster", db, out, tmp, "-c"],
                 ["cluster", db, out, tmp, "--frobnicate", "3"]):
        r = subprocess.run([shim] + argv, capture_output=True, text=True)
        assert r.returncode == 2, (argv, r.returncode, r.stderr)


def test_shim_answers_the_weight_download_call(tmp_path):
    """`foldseek databases ProstT5 <model> <model>/tmp --threads T` (createdb.rs:149-155) is what an unmodified Unicore spawns when
    <model>/prostt5-f16.gguf is missing: the shim cannot download, so it says where the file has to go and exits non-zero (Unicore then
    stops with its usual 'Command exited with code' message); with the file in place the call succeeds"""
    shim = os.path.join(ROOT, "bin", "foldseek")
    model = tmp_path / "weights"
    (model / "tmp").mkdir(parents=True)
    r = subprocess.run([shim, "databases", "ProstT5", str(model), str(model / "tmp"), "--threads", "4"], capture_output=True, text=True)
    assert r.returncode == 1 and "prostt5-f16.gguf" in r.stderr and "does not download" in r.stderr
    (model / "prostt5-f16.gguf").write_bytes(b"GGUF")
    r = subprocess.run([shim, "databases", "ProstT5", str(model), str(model / "tmp"), "--threads", "4"], capture_output=True, text=True)
    assert r.returncode == 0
    r = subprocess.run([shim, "databases", "PDB", str(model), str(model / "tmp")], capture_output=True, text=True)
    assert r.returncode == 2


def test_synthetic_matrix_needs_an_explicit_opt_in(tmp_path):
    """the shipped 3Di matrix is a seeded stand-in: without UC_ALLOW_SYNTHETIC=1 (or a real mat3di.out /
    --mat3di) the engine refuses to run instead of clustering with a meaningless matrix"""
    shim = os.path.join(ROOT, "bin", "foldseek")
    env = {k: v for k, v in os.environ.items() if k != "UC_ALLOW_SYNTHETIC"}
    r = subprocess.run([shim, "cluster", "db", "out", str(tmp_path / "t")], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "UC_ALLOW_SYNTHETIC=1" in r.stderr
    r = subprocess.run([shim, "version"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "synthetic stand-in" in r.stdout
    mat = os.path.join(ROOT, "unicore_amd", "data", "mat3di_synthetic.out")
    r = subprocess.run([shim, "cluster", "db", "out", str(tmp_path / "t"), "--mat3di", mat], capture_output=True, text=True, env=env)
    assert r.returncode in (3, 4)          # an explicit matrix is accepted; the run then fails on the missing DB / device


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus N` with no WORLD_SIZE (the form the driver uses for its BENCH run) must become a torch.distributed.run launch
    with N ranks instead of exiting with a usage error.  Without a GPU every rank then stops with the engine's message — what
    matters here is that N ranks were started."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_multi_gpu.py runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert (r.stdout + r.stderr).count("bench.py needs a GPU") == 2


def test_gemm_supply_microbenchmark_generator_still_matches_the_kernel_text(tmp_path):
    """tools/ubench/make_gemm_supply.py cuts the library's GEMM kernel out of uc_t5_kernels.hip by text anchors and switches parts of it off: every
    anchor must still be there (the generator asserts each substitution), and the committed gemm_supply.hip must be what it generates"""
    gen = os.path.join(ROOT, "tools", "ubench", "make_gemm_supply.py")
    committed = open(os.path.join(ROOT, "tools", "ubench", "gemm_supply.hip")).read()
    src = open(gen).read().replace('HERE = os.path.dirname(os.path.abspath(__file__))', 'HERE = %r' % os.path.join(ROOT, "tools", "ubench"))
    src = src.replace('open(os.path.join(HERE, "gemm_supply.hip"), "w")', 'open(%r, "w")' % str(tmp_path / "gemm_supply.hip"))
    exec(compile(src, gen, "exec"), {"__name__": "gen"})
    assert open(tmp_path / "gemm_supply.hip").read() == committed


def test_synthetic_generator_bytes_are_pinned_whatever_the_thread_count(tmp_path):
    """tools/gen_synth.c runs on threads since r06; every golden under tests/golden/ that was computed on a generated database depends on its BYTES.  The eight
    content files of two small databases, concatenated, hash to what the sequential generator of r01-r05 wrote (hashes taken with that binary), for 1, 3 and the
    default number of threads."""
    import hashlib
    gen = util.ensure_tools()
    want = {("3", "0x5EED0009", "60", "1.0"): "b7a0bbcd4d47a66caf0e0910e4608f102f7a8f521512b5a0efb5c5929f426f47",
            ("6", "0x5EED0004", "40", "0.6"): "5d2a87494c4c1769e038220b66453369425a062501d18f3c1abbed3d3c8f5843"}
    for args, sha in want.items():
        for threads in ("1", "3", None):
            prefix = str(tmp_path / ("db_%s_%s" % (args[0], threads)))
            env = dict(os.environ)
            env.pop("UC_GEN_THREADS", None)
            if threads:
                env["UC_GEN_THREADS"] = threads
            subprocess.check_call([gen, prefix] + list(args), env=env, stderr=subprocess.DEVNULL)
            h = hashlib.sha256()
            for sfx in ("", "_ss", "_h", ".index", "_ss.index", "_h.index", ".lookup", ".map"):
                h.update(open(prefix + sfx, "rb").read())
            assert h.hexdigest() == sha, (args, threads)


def test_bench_slim_line_of_a_committed_full_record():
    """bench.py's ONE line: the slim form of the committed full record of the round's driver-form run stays below 8 kB, carries the contract
    keys and the c3 / c4 figures at the top level, and loses nothing the headline needs (pure host logic: no GPU)."""
    import importlib.util
    import json
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    full = json.load(open(os.path.join(ROOT, "profiles", "r06", "r06f_bench_detail.json")))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        line = b.slim_line(full, types.SimpleNamespace(full_line=False, workdir=d, detail_dir=d, config="c2"))
        text = json.dumps(line)
        assert len(text) < 8000
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                  "cpu_baseline", "value_disk_to_tsv_aln_s", "value_one_shot_aln_s", "c3_value", "c3_ms_per_step", "c3_cpu_ratio", "c4_value", "c4_wall_s", "c5_mini_tflops", "detail_file"):
            assert k in line, k
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu_frac"):
            assert k in line["roofline"], k
        assert abs(line["value"] - full["value"]) <= 1e-5 * full["value"] and abs(line["c3_value"] - full["configs"]["c3"]["value"]) <= 1e-5 * line["c3_value"]
        assert set(line["configs"]) == {"c3", "c5-mini", "c4"} and "errors" not in line
        assert json.load(open(line["detail_file"]))["value"] == full["value"]
        # a failed sub-record stays visible, the headline stands
        broken = dict(full, configs=dict(full["configs"], c4={"error": "RuntimeError: out of memory"}))
        l2 = b.slim_line(broken, types.SimpleNamespace(full_line=False, workdir=d, detail_dir=d, config="c2"))
        assert l2["errors"] == {"c4": "RuntimeError: out of memory"} and "c4_value" not in l2 and l2["value"] == line["value"] and "c3_value" in l2
