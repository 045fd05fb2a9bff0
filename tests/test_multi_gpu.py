"""N > 1 on DISTINCT devices through real RCCL (the call that must use them: cluster.rs:45-49).  These tests
enable themselves on a box with >= 2 visible GPUs — the single-GPU test box skips them, the virtual-rank tests of
test_cli_gpu.py cover the same code minus RCCL there — so that the first time an 8-GPU node runs the suite nothing can fail
for a trivial reason: uc_cluster --gpus {2,4,8} (T = N, Q2 x T2, the multi-round exchange) against the 1-GPU TSV, RCCL's own
rank count, a rank that dies mid-run, and `python bench.py --gpus 2` launching itself under torch.distributed.run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
ROOT = util.ROOT


def _ndev():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


needs2 = pytest.mark.skipif(_ndev() < 2, reason="needs >= 2 visible GPUs (real RCCL ranks); virtual ranks are covered in test_cli_gpu.py")


@pytest.fixture(scope="module")
def db(tmp_path_factory):
    d = tmp_path_factory.mktemp("mgpu")
    return util.gen_synth_db(str(d / "db"), 8, 0x5EED0007, 60, 0.7)


def _run(db, out, opts, num_gpus, env=None):
    import unicore_amd as U
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        st = U.cluster(db, out + "_cluster", out + "_tmp", opts, threads=4, num_gpus=num_gpus)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    U.createtsv(db, out + "_cluster", out + ".tsv")
    return open(out + ".tsv", "rb").read(), st


@needs2
@pytest.mark.timeout(900)
@pytest.mark.parametrize("workflow", ["-c 0.8 --single-step-clustering", "-c 0.8"])
def test_real_rccl_ranks_give_the_one_gpu_tsv(db, workflow, tmp_path):
    ref, st1 = _run(db, str(tmp_path / "g1"), workflow, 1)
    nd = _ndev()
    cases = [(n, "") for n in (2, 4, 8) if n <= nd]
    if nd >= 4:
        cases.append((4, " --target-shards 2"))                     # Q2 x T2
    for n, extra in cases:
        for env in ({}, {"UC_EXCHANGE_LIMIT": "10"}):                # one-shot and multi-round exchange 1
            got, st = _run(db, str(tmp_path / ("g%d" % n)), workflow + extra, n, env=env)
            assert got == ref, (n, extra, env)
            assert st["n_gpus"] == n and st["nccl_ranks"] == n      # what ncclCommCount reports
            assert st["n_gapped_alignments"] == st1["n_gapped_alignments"] and st["n_clusters"] == st1["n_clusters"]
            assert st["exchange_bytes"] > 0 and sum(st["phase_seconds"]) > 0


@needs2
@pytest.mark.timeout(300)
@pytest.mark.parametrize("stage", [0, 1, 2])      # 2 = INSIDE the grouped point-to-point exchange (abort must not wait for the enqueue)
def test_a_rank_that_dies_gives_an_error_not_a_hang_real_rccl(db, stage, tmp_path):
    import unicore_amd as U
    with pytest.raises(U.UcError) as ei:
        _run(db, str(tmp_path / "dead"), "-c 0.8 --single-step-clustering", 2, env={"UC_FAIL_RANK": "1:%d" % stage})
    assert "GPU rank 1" in str(ei.value) and "injected" in str(ei.value)
    got, _ = _run(db, str(tmp_path / "again"), "-c 0.8 --single-step-clustering", 2)      # the process is still usable afterwards
    ref, _ = _run(db, str(tmp_path / "g1"), "-c 0.8 --single-step-clustering", 1)
    assert got == ref


@pytest.mark.timeout(300)
@pytest.mark.parametrize("stage", [0, 1])
def test_a_rank_that_dies_gives_an_error_not_a_hang_virtual_ranks(db, stage, tmp_path):
    """a rank that throws (engine construction, or between its prefilter and the exchange) must surface as an error
    code of uc_cluster — the survivors leave their next barrier instead of waiting for it forever"""
    import unicore_amd as U
    for n in (2, 3):
        with pytest.raises(U.UcError) as ei:
            _run(db, str(tmp_path / "dead"), "-c 0.8 --single-step-clustering", n, env={"UC_VIRTUAL_GPUS": "1", "UC_FAIL_RANK": "1:%d" % stage})
        assert ei.value.code == U.UC_ERR_DEVICE and "GPU rank 1" in str(ei.value) and "injected" in str(ei.value)


@pytest.mark.parametrize("opts", ["-c 0.8 --single-step-clustering", "-c 0.8 --single-step-clustering --min-seq-id 0.3 -s 6", "-c 0.8"])
def test_symmetric_rank_layout_equals_the_full_grid(db, opts, tmp_path):
    """T = N under a symmetric matrix (r04): a rank matches its target shard against its own sequences and the queries of the next N/2 shards only,
    the pairs the other way round are mirrored and travel with exchange 1 — same clust.tsv, same alignments, and the SAME algorithmic counters as
    the full shard x all-queries grid (UC_PREFILTER_SYMMETRIC=0) and as one GPU, for even and odd N, with the shards cut into index chunks too"""
    ref, st1 = _run(db, str(tmp_path / "g1"), opts, 1)
    for n in (2, 3, 4, 5, 8):
        for extra in ({}, {"UC_PREFILTER_CHUNK_RES": "9000"}):
            got, st = _run(db, str(tmp_path / ("s%d" % n)), opts, n, env=dict(extra, UC_VIRTUAL_GPUS="1"))
            full, stf = _run(db, str(tmp_path / ("f%d" % n)), opts, n, env=dict(extra, UC_VIRTUAL_GPUS="1", UC_PREFILTER_SYMMETRIC="0"))
            assert got == ref and full == ref, (n, extra)
            for k in ("n_gapped_alignments", "n_clusters", "n_kmer_hits", "n_candidates"):
                assert st[k] == st1[k] == stf[k], (n, extra, k, st[k], st1[k], stf[k])
            if "single-step" in opts and not extra:
                assert st["n_filtered_hits"] < stf["n_filtered_hits"], n        # fewer keys expanded and sorted


def test_phase_times_and_serialized_virtual_ranks(db, tmp_path):
    """the per-phase clock of the sharded pass (uc_stats.phase_seconds) and the emulation mode behind tools/critical_path.py:
    with UC_VIRTUAL_SERIAL=1 the compute phases of the virtual ranks take turns on the GPU; results are unchanged"""
    import unicore_amd as U
    ref, _ = _run(db, str(tmp_path / "g1"), "-c 0.8 --single-step-clustering", 1)
    got, st = _run(db, str(tmp_path / "g4"), "-c 0.8 --single-step-clustering", 4, env={"UC_VIRTUAL_GPUS": "1", "UC_VIRTUAL_SERIAL": "1"})
    assert got == ref and st["nccl_ranks"] == 0
    ph = dict(zip(U.PHASES, st["phase_seconds"]))
    assert ph["prefilter"] > 0 and ph["gapped"] > 0 and ph["merge_at_home"] > 0 and ph["install_owned"] > 0


@needs2
@pytest.mark.timeout(900)
def test_bench_launches_itself_for_n_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE re-executes itself under torch.distributed.run (the driver's BENCH form
    with N > 1 must not exit with a usage error): same alignments per step as one GPU, RCCL rank count in the line"""
    env = dict(os.environ, UC_BENCH_DIR=str(tmp_path / "bench"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--proteomes", "8", "--families", "60", "--len-scale", "0.7",
            "--no-cpu-baseline", "--no-extra-legs", "--no-sub-records"]
    lines = {}
    for n in (1, 2):
        r = subprocess.run(base + ["--gpus", str(n)], env=env, capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, r.stderr[-3000:]
        lines[n] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert lines[2]["n_gpus"] == 2 and lines[2]["rccl_ranks"] == 2
    assert lines[2]["config"]["alignments_per_step"] == lines[1]["config"]["alignments_per_step"]
    assert lines[2]["config"]["clusters"] == lines[1]["config"]["clusters"]


@pytest.mark.timeout(900)
def test_bench_n_gt_1_code_path_with_one_real_rccl_rank(tmp_path):
    """bench.py's N > 1 code path END TO END on the single-GPU box (VERDICT r04 item 6 iii): launched exactly as the driver launches N > 1 —
    `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1` — with UC_BENCH_FORCE_MULTI=1, so that one rank goes through
    the real ncclCommInitRank, the gloo control plane, the max / sum reductions over ranks and the configs[2] leg over all ranks (both
    configurations cut to 5 proteomes by the UC_BENCH_PROTEOMES test hook).  The line must carry every key an N-GPU line carries, and the
    same alignments and clusters as the plain single-GPU form of the same two workloads."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, UC_BENCH_DIR=str(tmp_path / "bench"), UC_BENCH_PROTEOMES="c2=5,c3=6", UC_ALLOW_SYNTHETIC="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "UC_BENCH_FORCE_MULTI"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable] + tail + ["--no-extra-legs", "--no-sub-records"], env=env, capture_output=True, text=True, timeout=800)
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail,
                       env=dict(env, UC_BENCH_FORCE_MULTI="1"), capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["scaling"] == "strong"
    assert "ragged all-to-all" in line["config"]["parallelism"]
    import unicore_amd as U
    assert set(line["phases_max_over_ranks_s_per_step"]) == set(U.PHASES)
    assert line["exchange_bytes_per_step_all_ranks"] >= 0 and line["exchange_rank0_s_per_step"] >= 0
    assert line["config"]["alignments_per_step"] == one["config"]["alignments_per_step"] and line["config"]["clusters"] == one["config"]["clusters"]
    c3 = line["configs"]["c3"]                                # the configs[2] leg of an N > 1 line
    assert c3["n_gpus"] == 1 and c3["rccl_ranks"] == 1 and c3["config"]["alignments_per_step"] > 0 and "phases_max_over_ranks_s_per_step" in c3
    assert "6 synthetic proteomes" in c3["config"]["workload"] and "5 synthetic proteomes" in line["config"]["workload"]


@pytest.mark.timeout(900)
def test_bench_c5_n_gt_1_code_path_with_one_real_rccl_rank(tmp_path):
    """`bench.py --config c5 --gpus N` (BASELINE configs[4] on N GPUs: one ProstT5 encoder replica per rank, every N-th sequence of the length-sorted order,
    codes gathered over the control plane, then the RCCL-sharded cluster step) through its N > 1 code path with ONE rank, launched as the driver launches
    N > 1; same alignments and clusters as the plain single-GPU form of the same workload (one synthetic proteome, the shared full-depth synthetic model)."""
    import socket
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_t5_full_depth as F
    gguf = F.ensure_gguf()
    work = os.path.dirname(gguf)
    assert os.path.basename(gguf) == "prostt5_synth_24.gguf"          # the file bench_c5 looks for in its --workdir
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, UC_ALLOW_SYNTHETIC="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "UC_BENCH_FORCE_MULTI"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--config", "c5", "--proteomes", "1", "--gpus", "1", "--steps", "1", "--warmup", "0", "--workdir", work]
    r1 = subprocess.run([sys.executable] + tail, env=env, capture_output=True, text=True, timeout=800)
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail,
                       env=dict(env, UC_BENCH_FORCE_MULTI="1"), capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and "encoder replicas" in line["config"]["parallelism"]
    assert line["config"]["alignments_per_step"] == one["config"]["alignments_per_step"] > 0 and line["config"]["clusters"] == one["config"]["clusters"]
    assert line["roofline"]["bound"] == "mfma" and 0.1 < line["roofline"]["frac"] < 1.0 and set(line["stages_s_per_step"]) == {"prostt5_encode", "gather_codes", "set_db_and_cluster"}


_TWO_RANK_SCRIPT = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import numpy as np
import torch                      # one HIP runtime per process (conftest.py)
import unicore_amd as U
rank, idf, db, out = int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
if rank == 0:
    uid = U.Comm.unique_id()
    open(idf + ".tmp", "wb").write(bytes(uid)); os.replace(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        if time.time() - t0 > 60: raise SystemExit("no id file")
        time.sleep(0.05)
    uid = open(idf, "rb").read()
c = U.Comm(uid, rank, 2, device=0)                      # BOTH ranks on device 0
print("COMM_OK", c.info(), flush=True)
e = U.Engine("-c 0.8", threads=2, verbosity=1, device=0)
e.load_db(db)
assign, n_aln = e.cluster_step(c, 0)
print("STEP_OK", n_aln, flush=True)
if rank == 0:
    np.save(out, assign)
c.close()
'''


@pytest.mark.timeout(300)
def test_two_real_rccl_ranks_on_one_device_if_rccl_permits(db, tmp_path):
    """the multi-rank branch of the RCCL exchange (grouped ncclSend / ncclRecv between DIFFERENT ranks, the device-to-device edge
    gather) has never run with world > 1 - the GPU box has one device.  Two PROCESSES, both on device 0, ask RCCL for a world-2 communicator.  If this ROCm's
    RCCL permits two ranks on one device (tried plain, then with the duplicate-GPU switches RCCL / NCCL builds have carried), the sharded step runs over it
    and its assignment must equal the single-rank one; if RCCL refuses (the documented behaviour: "Duplicate GPU detected"), the test SKIPS with RCCL's own
    words, so the record says what was tried.  Either way nothing hangs: both processes run under a timeout and are killed by PID."""
    import unicore_amd as U
    script = tmp_path / "two_rank.py"
    script.write_text(_TWO_RANK_SCRIPT)
    tried = []
    for extra in ({}, {"NCCL_IGNORE_DUPLICATE_GPU": "1", "RCCL_IGNORE_DUPLICATE_GPU": "1", "NCCL_ALLOW_DUPLICATE_GPU": "1", "RCCL_ALLOW_DUPLICATE_GPU": "1"}):
        idf, out = str(tmp_path / ("id_%d" % len(tried))), str(tmp_path / ("assign_%d.npy" % len(tried)))
        env = dict(os.environ, UC_ALLOW_SYNTHETIC="1", NCCL_DEBUG="WARN", **extra)
        procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), idf, db, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
        outs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=90)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
                o = (o or "") + "\n[killed after 90 s]"
            outs.append(o or "")
        if all(p.returncode == 0 for p in procs) and all("STEP_OK" in o for o in outs):
            e = U.Engine("-c 0.8", threads=2, verbosity=1)
            e.load_db(db)
            ref, _ = e.cluster_step()
            e.close()
            assert np.array_equal(np.load(out), ref), "two RCCL ranks on one device: assignment differs from one rank"
            return
        lines = [l.strip() for o in outs for l in o.splitlines()]
        noise = ("iommu=pt", "Could not read node")                         # RCCL start-up warnings of this container, not the refusal
        msg = [l for l in lines if "uplicate" in l] or [l for l in lines if any(w in l for w in ("NCCL WARN", "UcError", "error", "Error", "killed")) and not any(z in l for z in noise)]
        msg = [l.split("] ")[-1][-200:] for l in msg]
        tried.append("%s -> %s" % ("plain" if not extra else "with " + "/".join(sorted(extra)), " ; ".join(dict.fromkeys(msg[:3])) or "rc %s" % [p.returncode for p in procs]))
    pytest.skip("RCCL of this ROCm refuses two ranks on one device: " + " | ".join(tried)[:900])
