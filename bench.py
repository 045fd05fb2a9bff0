#!/usr/bin/env python3
"""bench.py — headline benchmark of the `unicore cluster` hot path on MI355X.

metric  : 3Di alignments/sec on the cluster path (BASELINE.json) = gapped 3Di+AA alignments handed to stage E5 divided
          by wall time.
step    : one full pass of the all-vs-all path (k-mer index -> similar-k-mer match -> ungapped -> top-M -> 3-pass gapped
          SW -> coverage/E-value gate -> set cover) through ONE C entry point, uc_engine_cluster_step, with the sequence
          DB already resident in HBM when the timed region starts.  `value` is that rate.
also    : `value_disk_to_tsv` — SURVEY.md 8(d)'s definition: wall of uc_cluster + uc_createtsv, DB on disk (page cache
          warm) -> clust.tsv written, same workload, measured after the timed region (best and mean of 3);
          `workflow_default` — what a bare "-c 0.8" runs (Foldseek's default: linear-time pre-step + 3-step cascade).
workload: --config c2 (default) = BASELINE.json configs[1]: 50 synthetic proteomes (~150 k sequences), tools/gen_synth.c
          seed 0x5EED0002, "-c 0.8", the plain all-vs-all step (`--single-step-clustering` semantics);
          --config c3 = configs[2] (500 proteomes, seed 0x5EED0003); --config c4-lite = configs[3]'s options
          ("-c 0.8 --min-seq-id 0.3 -s 7.5") on 50 proteomes (seed 0x5EED0004); --config c4 = configs[3] at its NOMINAL 2000
          proteomes through the DEFAULT workflow; --config c5 = configs[4]: the ProstT5 AA -> 3Di
          encoder (MFMA) fused ahead of the cluster path on --proteomes 5 (its own metric line, see bench_c5).
workflow: --workflow default (implied by c4; any config): one step = one uc_cluster call = what cluster.rs:45-49 triggers — the
          linear-time pre-step + 3-step cascade — with per-round records from the workflow observer and a CPU baseline that re-does
          the same rounds on the CPU (see bench_workflow / cpu_baseline_workflow).
N > 1   : one process per GPU (torch.distributed.run).  The data path is inside the library: the target DB is range-
          partitioned across the ranks (Q x T grid, T = N by default = the north-star layout), the per-shard hit lists go to the
          query's home rank with RCCL over xGMI from C (uc_comm_*: ragged all-to-all), are merged there, the surviving pairs go to
          their owner rank, every rank aligns the pairs it owns, edges go to rank 0 for the host set cover.  torch.distributed (gloo) only carries the 128-byte RCCL id, the
          barriers and the max-over-ranks of the timing.  Total work is fixed -> "strong".

          `python bench.py --gpus N` with WORLD_SIZE unset re-executes itself under torch.distributed.run with N ranks.
          N > 1 keeps configs[1] as the headline (the N = 1 point of a scaling run then equals the single-GPU bench) and adds
          `configs.c3`: ONE warm pass of configs[2] — the configuration BASELINE names for the 8-GPU node — over all N ranks, with
          per-phase maxima over ranks, rccl_ranks and exchange bytes.
also (N = 1, default config): `configs` — sub-records measured in the same run: c3 (BASELINE configs[2] = north_star's quoted
          1-GPU target size, 500 proteomes: ONE timed pass after one warm-up pass, with its own roofline block and CPU baseline sample), c5-mini
          (the ProstT5 encoder's MFMA fraction on one synthetic proteome) and c4 (BASELINE configs[3] at its NOMINAL 2000 proteomes: one
          uc_cluster call through the default workflow with its rooflines, rounds and round-by-round CPU leg; --no-c4 skips it); `value_one_shot_processes` — what an unmodified
          Unicore experiences: the two spawns of cluster.rs:45-64 (`foldseek cluster` + `foldseek createtsv` through the shim),
          wall from process start to clust.tsv.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")    # the benchmark runs on the seeded stand-in 3Di matrix (no real mat3di.out can be shipped)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # the engine's 8 streams need 8 hardware queues; read when HIP initialises

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0            # ... and the measured copy bandwidth of the same guide
VALU_PEAK_LANE_OPS = 39.3e12     # int32/packed VALU: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz — measured: every VALU instruction of the
                                 # SW kernel occupies its SIMD for 4 cycles (profiles/round1/r1b_pmc_sw.txt, tools/ubench/valu_rate.hip)
VALU_OPS_PER_CELL = 5.9          # static ISA count of the packed kernel's step loop at C2's class mix

CONFIGS = {
    # name: (proteomes, families, len_scale, seed, options, label)
    "c2": (50, 6000, 1.0, 0x5EED0002, "-c 0.8", "BASELINE configs[1]"),
    "c3": (500, 6000, 1.0, 0x5EED0003, "-c 0.8", "BASELINE configs[2] on the GPUs given"),
    # configs[3]'s options at the size one GPU finishes in half a minute: k-mer hits grow with the square of the database and
    # -s 7.5 already gives ~23x the hits of -s 4 (100 proteomes: 78 s per pass, 500: ~50 min; tools/c4_probe.py)
    "c4-lite": (50, 6000, 1.0, 0x5EED0004, "-c 0.8 --min-seq-id 0.3 -s 7.5", "BASELINE configs[3] options (-s 7.5, --min-seq-id 0.3) on 50 proteomes"),
    # configs[3] at its NOMINAL size.  Unicore forwards these options WITHOUT --single-step-clustering (cluster.rs:35,45-49), so what runs is the
    # default workflow (pre-step + 3-step cascade, the deep rounds on representatives only): `--config c4` therefore implies `--workflow default`
    "c4": (2000, 6000, 1.0, 0x5EED0004, "-c 0.8 --min-seq-id 0.3 -s 7.5", "BASELINE configs[3]"),
}


def pmc_json(name):
    f = os.path.join(ROOT, "profiles", name)
    return json.load(open(f)) if os.path.exists(f) else None


def gen_db(workdir, proteomes, families, scale, seed):
    os.makedirs(workdir, exist_ok=True)
    prefix = os.path.join(workdir, "db")
    if not os.path.exists(prefix + ".map"):
        gen = os.path.join(ROOT, "bin", "gen_synth")
        if not os.path.exists(gen):
            subprocess.check_call(["make", "-C", ROOT, "tools"], stdout=subprocess.DEVNULL)
        tmp = prefix + ".gen"
        subprocess.check_call([gen, tmp, str(proteomes), hex(seed), str(families), str(scale)], stderr=subprocess.DEVNULL)
        for sfx in ("", "_ss", "_h", ".index", "_ss.index", "_h.index", ".dbtype", "_ss.dbtype", "_h.dbtype", ".lookup", ".map"):
            os.replace(tmp + sfx, prefix + sfx)   # .map last: its presence marks a complete DB
    return prefix


def read_lens(prefix):
    idx = np.loadtxt(prefix + ".index", dtype=np.int64, ndmin=2)
    idx = idx[np.argsort(idx[:, 0], kind="stable")]
    return (idx[:, 2] - 2).astype(np.int64)


def real_foldseek():
    """a real Foldseek on PATH (never this repo's shim) — SURVEY.md 8(d) asks for it to be timed if present"""
    for d in os.environ.get("PATH", "").split(os.pathsep):
        f = os.path.join(d, "foldseek")
        if os.path.isfile(f) and os.access(f, os.X_OK) and os.path.realpath(f) != os.path.realpath(os.path.join(ROOT, "bin", "foldseek")):
            try:
                v = subprocess.run([f, "version"], capture_output=True, text=True, timeout=20).stdout.strip()
            except Exception:
                continue
            if "unicore-cluster" not in v:
                return f, v
    return None, None


def usable_cores():
    """threads worth starting: the affinity mask, capped by the cgroup CPU quota if there is one (a container that sees 256 CPUs
    may be allowed a fraction of them)"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                                   # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except Exception:
        try:                                               # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = max(1, int(q / per + 0.5))
        except Exception:
            pass
    return (min(cores, quota) if quota else cores), cores, quota


def cpu_baseline(prefix, opts, n_seqs, target_seconds=20.0):
    """CPU legs timed on this host's cores, on a bounded random query sample of the same workload (E2-E6 of the sampled
    queries against the FULL k-mer index; index build charged pro rata):
      kind "port": the oracle — the plain-C restatement of the spec (scalar DP), OpenMP, dynamic schedule over (query,
                   target) pairs;
      kind "simd": the same pipeline with the gapped stage as inter-sequence SIMD (16 targets of one query per AVX2
                   register, int16, query profile) — what a competent CPU implementation does (oracle/uc_simd.c).
    Both process the identical pair list, so alignments/s compare one to one with the GPU figure.  The SIMD leg also reports
    its gapped stage in GCUPS: one thread alone (the per-core figure a CPU person would quote) and all threads together —
    their ratio is the number of cores the box really gives this process (hyperthreads, quotas and neighbours included)."""
    from oracle import oracle_py as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    cores, affinity, quota = usable_cores()
    p = util.oracle_params(O, opts)
    odb = O.OracleDb(prefix)
    t0 = time.time()
    ix = O.build_index(odb, p)
    t_index = time.time() - t0
    order = np.random.default_rng(12345).permutation(n_seqs).astype(np.uint32)
    fs_path, fs_version = real_foldseek()
    out = []
    for kind in ("port", "simd"):
        if kind == "simd" and not hasattr(O, "simd_sample_run"):
            continue
        run = O.sample_run if kind == "port" else O.simd_sample_run
        extra = {}
        if kind == "simd":   # one thread alone: GCUPS per core of the gapped stage
            O.simd_cells()
            n0 = min(n_seqs, 48)
            _, _, ta0 = run(odb, ix, p, order[n_seqs - n0:], threads=1)
            c0 = O.simd_cells()
            extra["gapped_gcups_one_thread"] = c0[0] / ta0 / 1e9 if ta0 > 0 else 0.0
        n1 = min(n_seqs, max(64, 2 * cores))                               # calibration sample
        a1, tp1, ta1 = run(odb, ix, p, order[:n1], threads=cores)
        rate = (tp1 + ta1) / max(n1, 1)
        # fill the time budget, but never fewer than 20 queries per thread (a smaller sample is tail-dominated)
        n2 = int(max(0, (target_seconds - (tp1 + ta1)) / max(rate, 1e-9)))
        n2 = max(n2, 20 * cores - n1)
        n2 = max(0, min(n2, n_seqs - n1))
        a2, tp2, ta2 = (0, 0.0, 0.0)
        if kind == "simd":
            O.simd_cells()
        if n2 > 0:
            a2, tp2, ta2 = run(odb, ix, p, order[n1:n1 + n2], threads=cores)
        # rate from the large sample alone (the calibration run also pays thread start-up and cold caches)
        nq, aln, tp, ta = (n2, a2, tp2, ta2) if n2 > 0 else (n1, a1, tp1, ta1)
        if kind == "simd" and n2 > 0:
            cu, cs = O.simd_cells()
            extra["gapped_gcups_all_threads"] = cu / ta / 1e9 if ta > 0 else 0.0
            extra["gapped_cells"] = cu
            extra["lane_fill"] = cu / cs if cs else 0.0
            g1 = extra.get("gapped_gcups_one_thread", 0.0)
            extra["effective_cores"] = extra["gapped_gcups_all_threads"] / g1 if g1 > 0 else None
        t = tp + ta + t_index * nq / n_seqs
        out.append(dict({"value": aln / t if t > 0 else 0.0, "unit": "alignments/s", "cores": cores, "kind": kind,
                    "threads_started": cores, "cpus_in_affinity_mask": affinity, "cgroup_cpu_quota": quota,
                    "implementation": ("oracle/uc_oracle.c (scalar C restatement of the spec, gcc -O3 -march=x86-64-v3, OpenMP dynamic)" if kind == "port" else
                                       "oracle/uc_simd.c (AVX2 inter-sequence int16 Smith-Waterman, 16 targets per register, + the oracle's prefilter; OpenMP dynamic)"),
                    "foldseek": ("%s (%s)" % (fs_path, fs_version)) if fs_path else "not available — the CPU baseline is this repository's own code, not Foldseek",
                    "sample": "%d of %d queries (random, seed 12345, %.1f per thread): %d gapped alignments; prefilter %.2fs + gapped %.2fs "
                              "+ pro-rata index build %.3fs of %.2fs" % (nq, n_seqs, nq / cores, aln, tp, ta, t_index * nq / n_seqs, t_index)}, **extra))
    O.free_index(ix)
    return out


def one_shot_processes(prefix, workdir, options, aln_plain, aln_default):
    """What an unmodified Unicore experiences (cluster.rs:45-64): a NEW `foldseek cluster` process and a NEW `foldseek createtsv`
    process per call, through the argv-compatible shim with the reference's token order; wall from the first spawn to clust.tsv
    on disk (process start, HIP initialisation, code-object load and first allocations included), page cache warm, best of 2."""
    shim = os.path.join(ROOT, "bin", "foldseek")
    outp = os.path.join(workdir, "oneshot_clust")
    tmp = os.path.join(workdir, "tmp")
    res = {"what": "bin/foldseek cluster + bin/foldseek createtsv as two fresh processes (cluster.rs:45-64 argv), first spawn -> clust.tsv; best of 2", "unit": "alignments/s"}
    T = str(usable_cores()[0])      # what `unicore --threads 0` would pass on a box whose quota it respects
    for tag, extra, aln in (("plain_step", ["--single-step-clustering"], aln_plain), ("default_workflow", [], aln_default)):
        walls = []
        for _ in range(2):
            t0 = time.perf_counter()
            subprocess.check_call([shim, "cluster", "--threads", T, "-v", "1", prefix, outp + "_cluster", tmp] + options.split() + extra, stdout=subprocess.DEVNULL)
            subprocess.check_call([shim, "createtsv", "--threads", T, "-v", "1", prefix, prefix, outp + "_cluster", outp + ".tsv"], stdout=subprocess.DEVNULL)
            walls.append(time.perf_counter() - t0)
        res[tag] = {"wall_s_best": min(walls), "wall_s_all": walls, "alignments": aln, "value": aln / min(walls)}
    subprocess.call([shim, "rmdb", outp + "_cluster", "-v", "1"], stdout=subprocess.DEVNULL)
    return res


def c5_mini(args):
    """sub-record of the default line: the ProstT5 AA -> 3Di encoder (BASELINE configs[4]'s MFMA stage) on ONE synthetic proteome —
    full ProtT5-XL geometry, all 24 blocks, seeded random-init f16 weights — for its fraction of the dense f16 MFMA peak."""
    import unicore_amd as U
    from oracle import prostt5_ref as R
    seed = 0x5EED0005
    workdir = os.path.join(args.workdir, "p1_f6000_s1_%x" % seed)
    prefix = gen_db(workdir, 1, 6000, 1.0, seed)
    aa = [e.decode() for e in open(prefix, "rb").read().split(b"\n\0")[:-1]]
    gguf = os.path.join(args.workdir, "prostt5_synth_24.gguf")
    if not os.path.exists(gguf):
        R.write_synthetic_gguf(gguf + ".tmp", R.default_config(), seed=seed)
        os.replace(gguf + ".tmp", gguf)
    enc = U.T5Encoder(gguf)
    enc.encode(aa[:64])                                   # warm-up: code objects, first allocations
    s0 = enc.stats()
    t0 = time.perf_counter()
    enc.encode(aa)
    dt = time.perf_counter() - t0
    s1 = enc.stats()
    enc.close()
    fl, ms = s1["flops"] - s0["flops"], s1["gpu_ms"] - s0["gpu_ms"]
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    res = sum(len(x) for x in aa)
    return {"note": "ONE proteome: configs[4] at its nominal 500 proteomes is ~1,190 s of encoder on one GPU at this rate (profiles/r05/c5_p500_check.json, builder-run) and fits "
                    "neither driver step; on N GPUs uc_createdb runs one encoder replica per GPU (sequences sharded, no collective): ~150 s expected at N = 8",
            "config": {"workload": "BASELINE configs[4]'s encoder stage on 1 synthetic proteome: %d sequences, %d residues; ProtT5-XL geometry, 24 blocks, seeded random-init f16 weights" % (len(aa), res)},
            "encoder_residues_per_s": res / dt, "wall_s": dt,
            "roofline": {"bound": "mfma", "kernel": "t5_gemm256_kernel + t5_attention_kernel (f16 v_mfma_f32_16x16x32_f16, fp32 accumulate)", "achieved": tf, "peak": 2500.0,
                         "unit": "TFLOP/s", "frac": tf / 2500.0, "traffic": None, "algorithmic_flops": fl, "gpu_ms": ms,
                         "note": "algorithmic FLOPs = linear layers (2 x tokens x weights) + attention (4 L^2 x 4096 per sequence and block) / HIP-event time of the encoder passes"}}


def bench_c5(args, world=1, rank=0, local_rank=0, multi=False):
    """BASELINE configs[4]: createdb's ProstT5 AA -> 3Di encoder (hand-written f16 MFMA kernels) fused ahead of the cluster
    path, no disk round trip: AA residues -> uc_t5_encode -> 3Di codes -> uc_engine_set_db -> uc_engine_cluster_step.
    ProtT5-XL geometry (24 blocks, 1024 / 32 x 128 / 16384) with seeded random-init weights (the real prostt5-f16.gguf
    cannot be shipped; same loader).  One step = the whole chain on `--proteomes` synthetic proteomes (default 5).
    N > 1 (r06; one process per GPU): the encoder shards by sequence - rank r encodes every N-th sequence of the length-sorted order ("replicas only",
    no data-path collective in the encoder: the codes are gathered over the gloo control plane, as a file system would carry them between `unicore
    createdb` and `unicore cluster`) - then every rank holds the whole 3Di database and the cluster step runs sharded over RCCL as for --config c2."""
    import torch
    import unicore_amd as U
    from oracle import prostt5_ref as R
    dist = comm = None
    rccl_ranks = 0
    if multi:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        uid = [U.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = U.Comm(uid[0], rank, world, device=local_rank)
        rccl_ranks = comm.info()[0]

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    proteomes = args.proteomes if args.proteomes is not None else 5
    seed = 0x5EED0005
    workdir = os.path.join(args.workdir, "p%d_f6000_s1_%x" % (proteomes, seed))
    gguf = os.path.join(args.workdir, "prostt5_synth_24.gguf")
    if rank == 0:
        gen_db(workdir, proteomes, 6000, 1.0, seed)
        if not os.path.exists(gguf):
            R.write_synthetic_gguf(gguf + ".tmp", R.default_config(), seed=seed)
            os.replace(gguf + ".tmp", gguf)
    barrier()
    prefix = os.path.join(workdir, "db")
    aa = [e.decode() for e in open(prefix, "rb").read().split(b"\n\0")[:-1]]
    n, residues = len(aa), sum(len(x) for x in aa)
    enc = U.T5Encoder(gguf, device=local_rank)
    options = args.options if args.options is not None else "-c 0.8"
    eng = U.Engine(options, threads=max(1, (os.cpu_count() or 1) // world), verbosity=1, device=local_rank)
    lut = np.full(256, 20, np.uint8)
    for i, c in enumerate("ACDEFGHIKLMNPQRSTVWY"):
        lut[ord(c)] = i
    sa = lut[np.frombuffer("".join(aa).encode(), np.uint8)]
    lens = np.array([len(x) for x in aa], np.int64)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    order = np.argsort(-lens, kind="stable")
    mine = order[rank::world]                              # every N-th sequence of the length-sorted order: equal residue counts AND equal length mixes per rank
    t_enc = t_gather = t_clu = 0.0
    n_aln = 0
    assign = None

    def step(timed):
        nonlocal t_enc, t_gather, t_clu, n_aln, assign
        t0 = time.perf_counter()
        codes = enc.encode([aa[i] for i in mine])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s3 = np.empty(residues, np.uint8)
        if multi:
            parts = [None] * world
            dist.all_gather_object(parts, np.concatenate(codes) if len(codes) else np.zeros(0, np.uint8))
            for r in range(world):
                ids = order[r::world]
                o = 0
                for i in ids:
                    s3[int(off[i]):int(off[i + 1])] = parts[r][o:o + lens[i]]
                    o += lens[i]
        else:
            for k, i in enumerate(mine):
                s3[int(off[i]):int(off[i + 1])] = codes[k]
        t2 = time.perf_counter()
        eng.set_db(off, s3, sa)
        assign, a = eng.cluster_step(comm, args.target_shards) if multi else eng.cluster_step()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if timed:
            t_enc += t1 - t0; t_gather += t2 - t1; t_clu += t3 - t2; n_aln += a
    for _ in range(args.warmup):
        step(False)
    s0 = enc.stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    s1 = enc.stats()
    steps = max(args.steps, 1)
    fl, ms = s1["flops"] - s0["flops"], s1["gpu_ms"] - s0["gpu_ms"]
    if multi:
        v = torch.tensor([dt, t_enc, t_gather, t_clu, ms], dtype=torch.float64)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        dt, t_enc, t_gather, t_clu, ms = [float(x) for x in v]
        c = torch.tensor([float(n_aln), fl], dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        n_aln, fl = int(c[0].item()), float(c[1].item())
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0            # all replicas together: summed FLOPs over the slowest replica's GPU time
    if rank == 0:
        print(json.dumps({
            "metric": "3Di alignments/sec (createdb ProstT5 AA->3Di encoder + cluster path, end to end)", "value": n_aln / dt, "unit": "alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16 (MFMA, fp32 accumulate) + u16 (packed DP)", "data": "synthetic (sequences and random-init weights)",
            "config": {"workload": "BASELINE configs[4] chain at %d synthetic proteomes: %d sequences, %d residues; ProtT5-XL-geometry encoder (24 blocks, d_model 1024, "
                                   "32 x 128 heads, d_ff 16384, 3Di CNN head) with seeded random-init f16 weights -> 3Di codes -> cluster '%s' (plain all-vs-all step); "
                                   "no disk round trip" % (proteomes, n, residues, options),
                       "alignments_per_step": n_aln // steps, "clusters": int((assign == np.arange(n)).sum()) if assign is not None else None,
                       "parallelism": ("%d encoder replicas (one per GPU, every N-th sequence of the length-sorted order, no data-path collective), then the cluster step "
                                       "sharded over RCCL" % world) if multi else "single GPU"},
            "rccl_ranks": rccl_ranks,
            "stages_s_per_step": {"prostt5_encode": t_enc / steps, "gather_codes": t_gather / steps, "set_db_and_cluster": t_clu / steps},
            "encoder_residues_per_s": residues * steps / t_enc if t_enc > 0 else 0.0,
            "roofline": {"bound": "mfma", "kernel": "t5_gemm_kernel + t5_attention_kernel (f16 v_mfma_f32_16x16x32_f16, fp32 accumulate)", "achieved": tf, "peak": 2500.0 * world,
                         "unit": "TFLOP/s", "frac": tf / (2500.0 * world), "traffic": None,
                         "algorithmic_flops_per_step": fl / steps, "gpu_ms_per_step": ms / steps,
                         "note": "algorithmic FLOPs = linear layers (2 x tokens x weights) + attention (4 L^2 x 4096 per sequence and block), summed over the replicas / HIP-event "
                                 "time of the slowest replica's encoder passes; peak = 2.5 PFLOP/s dense f16 per GPU"},
        }))
    if multi:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()


def cpu_baseline_workflow(prefix, options, rounds, total_alignments, seconds_per_round=8.0):
    """CPU leg of the DEFAULT workflow on the same pair lists: the rounds the GPU run went through (their sequence sets and k-mer
    thresholds, from the workflow observer) are re-done round by round with the SIMD CPU code on a bounded random query sample against the
    round's FULL index and scaled to the round's size (the per-round hit lists equal the GPU's: tests/test_workflow_gpu.py); the index builds
    and the pre-step's candidate generation are timed in full.  value = the run's alignments / the sum of the round estimates."""
    from oracle import oracle_py as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    cores, affinity, quota = usable_cores()
    odb = O.OracleDb(prefix)
    est_total, per_round = 0.0, []
    for r in rounds:
        sub = odb.subset(r["ids"])
        m = sub.n
        p = util.oracle_params(O, options)
        rng = np.random.default_rng(4242 + r["round"])
        if r["round"] < 0:
            t0 = time.time()
            pairs = O.linclust_pairs(sub, p, 20)
            t_pairs = time.time() - t0
            centres = np.unique(pairs[:, 0])
            take = centres[rng.permutation(len(centres))[: max(64, 40 * cores)]] if len(centres) else centres
            lo, hi = np.searchsorted(pairs[:, 0], take, "left"), np.searchsorted(pairs[:, 0], take, "right")
            L = O.lib()
            import ctypes as C
            L.uco_simd_align_query.argtypes = [C.POINTER(O.Db), C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(O.Params), C.c_int32, C.c_void_p]
            L.uco_simd_align_query.restype = None
            done = 0
            t0 = time.time()                                    # one thread: scaled by the cores below (a query's list is the unit of parallel work)
            for k, q in enumerate(take):
                tg = np.ascontiguousarray(pairs[lo[k]:hi[k], 1], np.uint32)
                out = np.zeros(len(tg), O.ALN_DTYPE)
                L.uco_simd_align_query(C.byref(sub.db), int(q), tg.ctypes.data, len(tg), C.byref(p), O.min_score(sub, p, int(q)), out.ctypes.data)
                done += len(tg)
                if time.time() - t0 > seconds_per_round:
                    break
            t_s = time.time() - t0
            est = t_pairs + (t_s / max(done, 1)) * len(pairs) / cores
            per_round.append({"round": -1, "sequences": m, "pairs": int(len(pairs)), "candidate_generation_s": t_pairs,
                              "sampled_pairs": done, "estimated_s": est})
        else:
            p.kmer_thr = r["kmer_thr"]
            t0 = time.time()
            ix = O.build_index(sub, p)
            t_ix = time.time() - t0
            order = rng.permutation(m).astype(np.uint32)
            n1 = min(m, max(64, 2 * cores))
            a1, tp1, ta1 = O.simd_sample_run(sub, ix, p, order[:n1], threads=cores)
            rate = (tp1 + ta1) / max(n1, 1)
            n2 = max(0, min(m - n1, max(20 * cores - n1, int((seconds_per_round - tp1 - ta1) / max(rate, 1e-9)))))
            a2, tp2, ta2 = O.simd_sample_run(sub, ix, p, order[n1:n1 + n2], threads=cores) if n2 else (a1, tp1, ta1)
            nq = n2 if n2 else n1
            O.free_index(ix)
            est = t_ix + (tp2 + ta2) * m / max(nq, 1)
            per_round.append({"round": r["round"], "sequences": m, "kmer_thr": r["kmer_thr"], "index_build_s": t_ix, "sampled_queries": nq,
                              "sampled_alignments": a2, "sample_s": tp2 + ta2, "estimated_s": est})
        est_total += est
        del sub
    return {"value": total_alignments / est_total if est_total > 0 else 0.0, "unit": "alignments/s", "cores": cores, "kind": "simd",
            "threads_started": cores, "cpus_in_affinity_mask": affinity, "cgroup_cpu_quota": quota, "estimated_workflow_wall_s": est_total,
            "implementation": "oracle/uc_simd.c + the oracle's prefilter and pre-step (OpenMP), round by round on the sequence sets and thresholds of the GPU run",
            "foldseek": "not available — the CPU baseline is this repository's own code, not Foldseek",
            "sample": "per round: index build timed in full, a random query sample (~%.0f s) against the round's full index scaled to the round's size" % seconds_per_round,
            "rounds": per_round}


def bench_workflow(args, proteomes, families, len_scale, seed, options, label, custom, steps=None, warmup=None, cpu_seconds=None):
    """-> the line as a dict.  `--workflow default`: one step = one `uc_cluster(db, options)` call — the call of cluster.rs:45-49, which forwards the option string
    WITHOUT --single-step-clustering — i.e. the linear-time pre-step + the 3-step cascade, one C entry point, from the DB files (page cache
    warm) to the cluster DB.  `value` excludes the host-side read + encode of the DB files (uc_stats.stage_seconds[load]; the contract's
    'inputs resident'); `value_disk_to_cluster_db` includes it.  One GPU (the N-GPU form of the workflow runs inside uc_cluster with --gpus N)."""
    import torch
    import unicore_amd as U
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    workdir = os.path.join(args.workdir, "p%d_f%d_s%g_%x" % (proteomes, families, len_scale, seed))
    prefix = gen_db(workdir, proteomes, families, len_scale, seed)
    threads = usable_cores()[0]
    outp = os.path.join(workdir, "bench_wf")
    rounds, stamp = [], [0.0]

    def hook(rnd, ids, kthr, view):
        now = time.perf_counter()
        rounds.append({"round": rnd, "ids": ids, "kmer_thr": kthr, "pairs": view.hits_size(), "t_since_call_s": now - stamp[0]})

    def one(keep_rounds):
        rounds.clear()
        U.set_round_hook(hook if keep_rounds else None)
        stamp[0] = time.perf_counter()
        try:
            st = U.cluster(prefix, outp + "_cluster", os.path.join(workdir, "tmp"), options, threads=threads)
        finally:
            U.set_round_hook(None)
        return st, time.perf_counter() - stamp[0]
    n_steps = args.steps if steps is None else steps
    n_warm = args.warmup if warmup is None else warmup
    for _ in range(n_warm):
        one(False)
    walls, loads, sts = [], [], []
    for k in range(n_steps):
        st, w = one(k == n_steps - 1)
        walls.append(w); loads.append(st["stage_seconds"][0]); sts.append(st)
    st = sts[-1]
    steps_ = max(n_steps, 1)
    n_aln = sum(x["n_gapped_alignments"] for x in sts)
    dt, dl = sum(walls), sum(loads)
    sw_s = sum(x["sw_kernel_ms"] for x in sts) / 1e3
    pre_s = sum(x["prefilter_kernel_ms"] for x in sts) / 1e3
    cells_run = sum(x["cells_run"] for x in sts)
    cells_alg = sum(x["cells_fwd"] + x["cells_rev"] + x["cells_start"] + x["cells_tb"] for x in sts)     # the spec's four passes (oracle counts; cells_tb = the traceback boxes)
    ab = {k: sum(x["algorithmic_bytes"][i] for x in sts) for i, k in enumerate(U.STAGES)}
    pre_bytes = ab["index"] + ab["kmer"] + ab["ungapped"] + ab["select"]
    sw_bytes = sum(x["sw_algorithmic_bytes"] for x in sts)
    launches = sum(x["sw_kernel_launches"] for x in sts)
    prev = 0.0
    rr = []
    for r in rounds:
        rr.append({"round": r["round"], "sequences": int(len(r["ids"])), "kmer_thr": r["kmer_thr"], "pairs_aligned": int(r["pairs"]),
                   "s_until_gapped_stage_done": r["t_since_call_s"] - prev})
        prev = r["t_since_call_s"]
    out = {
        "metric": "3Di alignments/sec (cluster path)", "value": n_aln / (dt - dl), "unit": "alignments/s", "n_gpus": 1, "steps": n_steps, "warmup": n_warm,
        "ms_per_step": 1e3 * (dt - dl) / steps_, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": "%s: %d synthetic proteomes, %d seqs, %d residues, options '%s' as cluster.rs:45-49 forwards them (no --single-step-clustering): DEFAULT workflow = "
                               "linear-time pre-step + 3-step cascade, gen_synth seed %#x, synthetic stand-in 3Di matrix" % (label if not custom else "custom size", proteomes,
                                                                                                                        st["n_seqs"], st["n_residues"], options, seed),
                   "workflow": "default", "alignments_per_step": n_aln // steps_, "clusters": st["n_clusters"], "parallelism": "single GPU"},
        "value_definition": "one uc_cluster call per step (the foldseek-cluster spawn of cluster.rs:45-49) from the DB files to the cluster DB; value excludes the host-side read + encode "
                            "of the DB files (load_s_per_step), value_disk_to_cluster_db includes it",
        "value_disk_to_cluster_db": n_aln / dt, "wall_s_per_step": dt / steps_, "load_s_per_step": dl / steps_,
        "rounds_last_step": rr,
        "roofline": {"bound": "hbm", "kernel": "sw_pk_kernel + sw_group_kernel (gapped 3Di+AA SW, all classes, passes and rounds)",
                     "achieved": sw_bytes / sw_s / 1e9 if sw_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (sw_bytes / sw_s / 1e9 / HBM_PEAK_GBS) if sw_s > 0 else 0.0,
                     "traffic": None, "algorithmic_bytes_per_step": sw_bytes / steps_, "kernel_ms_per_step": 1e3 * sw_s / steps_,
                     "avg_launch_ms": 1e3 * sw_s / max(launches, 1), "launches": launches,
                     "note": "integer-VALU-bound by design (SURVEY.md 8d): read valu_frac",
                     "cells_run_per_step": cells_run / steps_, "cells_algorithmic_per_step": cells_alg / steps_,
                     "valu_peak_lane_ops": VALU_PEAK_LANE_OPS, "valu_peak_lane_ops_guide_nominal": 2 * VALU_PEAK_LANE_OPS, "valu_ops_per_cell": VALU_OPS_PER_CELL,
                     "valu_frac": (cells_run * VALU_OPS_PER_CELL / sw_s / VALU_PEAK_LANE_OPS) if sw_s > 0 else 0.0,
                     "valu_frac_of_guide_nominal_peak": (cells_run * VALU_OPS_PER_CELL / sw_s / (2 * VALU_PEAK_LANE_OPS)) if sw_s > 0 else 0.0},
        "roofline_prefilter": {"bound": "hbm", "kernels": "kmer_extract, sim_runs, filter, compact, diag_select, ungapped, select/rank/scatter + rocPRIM sorts (all cascade rounds)",
                               "algorithmic_bytes_per_step": pre_bytes / steps_, "kernel_ms_per_step": 1e3 * pre_s / steps_,
                               "achieved": pre_bytes / pre_s / 1e9 if pre_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": (pre_bytes / pre_s / 1e9 / HBM_PEAK_GBS) if pre_s > 0 else 0.0, "traffic_per_step": None},
        "algorithmic_bytes_per_step": {k: v / steps_ for k, v in ab.items()},
        "stages_s_per_step": {k: sum(x["stage_seconds"][i] for x in sts) / steps_ for i, k in enumerate(U.STAGES)},
        "prefilter_kernel_ms_per_step": 1e3 * pre_s / steps_, "sw_kernel_ms_per_step": 1e3 * sw_s / steps_,
        "counts_per_step": {k: sum(x[k] for x in sts) // steps_ for k in ("n_index_entries", "n_sim_kmers", "n_kmer_hits", "n_filtered_hits", "n_candidates", "n_prefilter_hits",
                                                                         "n_gapped_alignments", "n_start_alignments", "n_pk_reruns", "n_sw_runs")},
    }
    U.lib().uc_release_scratch()
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_workflow(prefix, options, rounds, st["n_gapped_alignments"],
                                                    seconds_per_round=max(2.0, (args.cpu_seconds if cpu_seconds is None else cpu_seconds) / 2.5))
    return out


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _slim_roof(r):
    return _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "valu_frac", "valu_frac_of_guide_nominal_peak", "valu_gcups", "kernel_ms_per_step",
                     "avg_launch_ms", "launches", "algorithmic_bytes_per_launch", "algorithmic_bytes_per_step", "cells_run_per_step", "cells_algorithmic_per_step",
                     "achieved_tflops", "algorithmic_flops", "gpu_ms", "traffic_per_step", "frac_of_measured_copy_bw"))


def _slim_cpu(c):
    o = _pick(c, ("value", "unit", "cores", "kind", "estimated_workflow_wall_s", "gapped_gcups_all_threads", "effective_cores"))
    if isinstance(c, dict) and "sample" in c:
        o["sample"] = c["sample"][:160]
    return o


def _slim_record(o):
    """a sub-record (or the headline) without its prose and per-round arrays: what a reader of the driver's 10 kB tail needs"""
    if not isinstance(o, dict) or "error" in o:
        return o
    r = _pick(o, ("value", "n_gpus", "rccl_ranks", "phases_max_over_ranks_s_per_step", "exchange_bytes_per_step_all_ranks", "ms_per_step", "steps", "warmup", "wall_s_per_step", "load_s_per_step", "value_disk_to_cluster_db", "encoder_residues_per_s", "wall_s",
                  "prefilter_kernel_ms_per_step", "sw_kernel_ms_per_step", "stages_s_per_step", "note"))
    if "config" in o:
        r["config"] = dict(_pick(o["config"], ("alignments_per_step", "clusters", "workflow")), workload=str(o["config"].get("workload", ""))[:110])
    if "roofline" in o:
        r["roofline"] = _pick(o["roofline"], ("bound", "achieved", "peak", "unit", "frac", "valu_frac", "kernel_ms_per_step", "launches", "cells_run_per_step", "gpu_ms"))
    if "roofline_prefilter" in o:
        r["roofline_prefilter"] = _pick(o["roofline_prefilter"], ("achieved", "frac", "kernel_ms_per_step", "algorithmic_bytes_per_step"))
    if "cpu_baseline" in o:
        r["cpu_baseline"] = _slim_cpu(o["cpu_baseline"])
    if "workflow_default" in o:
        r["workflow_default"] = _pick(o["workflow_default"], ("wall_s", "wall_s_best", "alignments", "clusters", "value"))
    if "rounds_last_step" in o:
        r["rounds"] = [[x["round"], x["sequences"], x["pairs_aligned"], round(x["s_until_gapped_stage_done"], 2)] for x in o["rounds_last_step"]]
        r["rounds_columns"] = "round (-1 = pre-step), sequences, pairs aligned, seconds"
    return r


def slim_line(out, args):
    """The ONE line the driver records (BENCH_r05.json kept a 10 kB tail of a 20 kB line and lost configs.c3.value).  The figures a
    reader needs sit at the TOP LEVEL and early; the full record (formulas, launch notes, per-round CPU arrays, both CPU legs) goes to a side file whose
    path the line carries.  --full-line prints the full record instead."""
    if out is None or args.full_line:
        return out
    detail = None
    for d in ((args.detail_dir,) if getattr(args, "detail_dir", None) else (os.path.join(ROOT, "gpurun_out"), args.workdir)):
        try:
            os.makedirs(d, exist_ok=True)
            detail = os.path.join(d, "bench_detail_%s_n%d.json" % (args.config, out.get("n_gpus", 1)))
            json.dump(out, open(detail, "w"), indent=1, default=lambda x: x.tolist() if hasattr(x, "tolist") else str(x))
            break
        except OSError:
            detail = None
    head = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    line = dict(head)
    cfg = out.get("config", {})
    line["config"] = dict(_pick(cfg, ("alignments_per_step", "clusters", "workflow")), workload=str(cfg.get("workload", ""))[:260], parallelism=str(cfg.get("parallelism", ""))[:200])
    line["value_definition"] = "value = alignments / wall of the timed steps with the DB resident in HBM (the bench contract); value_disk_to_tsv_aln_s = SURVEY.md 8(d)'s disk -> clust.tsv wall"
    top = {}
    v = out.get("value_disk_to_tsv")
    if isinstance(v, dict):
        top["value_disk_to_tsv_aln_s"] = v.get("value"); top["disk_to_tsv_wall_s"] = v.get("wall_s_best"); top["disk_to_tsv_same_clusters"] = v.get("same_clusters_as_steps")
    v = out.get("workflow_default")
    if isinstance(v, dict):
        top["workflow_default_wall_s"] = v.get("wall_s_best"); top["workflow_default_aln_s"] = v.get("value")
    v = out.get("value_one_shot_processes")
    if isinstance(v, dict):
        top["value_one_shot_aln_s"] = v.get("plain_step", {}).get("value"); top["one_shot_wall_s"] = v.get("plain_step", {}).get("wall_s_best")
        top["one_shot_default_workflow_wall_s"] = v.get("default_workflow", {}).get("wall_s_best")
    subs = out.get("configs", {}) or {}
    c3 = subs.get("c3")
    if isinstance(c3, dict) and "error" not in c3:
        top["c3_value"] = c3.get("value"); top["c3_ms_per_step"] = c3.get("ms_per_step")
        cb = c3.get("cpu_baseline", {})
        top["c3_cpu_value"] = cb.get("value"); top["c3_cpu_cores"] = cb.get("cores")
        top["c3_cpu_ratio"] = (c3["value"] / cb["value"]) if cb.get("value") else None
        top["c3_sw_kernel_s"] = c3.get("sw_kernel_ms_per_step", 0) / 1e3; top["c3_prefilter_kernel_s"] = c3.get("prefilter_kernel_ms_per_step", 0) / 1e3
        top["c3_valu_frac"] = c3.get("roofline", {}).get("valu_frac"); top["c3_prefilter_frac"] = c3.get("roofline_prefilter", {}).get("frac")
        top["c3_workflow_default_wall_s"] = c3.get("workflow_default", {}).get("wall_s")
    c4 = subs.get("c4")
    if isinstance(c4, dict) and "error" not in c4:
        top["c4_value"] = c4.get("value"); top["c4_wall_s"] = c4.get("wall_s_per_step"); top["c4_alignments"] = c4.get("config", {}).get("alignments_per_step")
        top["c4_sw_kernel_s"] = c4.get("sw_kernel_ms_per_step", 0) / 1e3; top["c4_prefilter_kernel_s"] = c4.get("prefilter_kernel_ms_per_step", 0) / 1e3
        top["c4_valu_frac"] = c4.get("roofline", {}).get("valu_frac"); top["c4_prefilter_frac"] = c4.get("roofline_prefilter", {}).get("frac")
        cb = c4.get("cpu_baseline", {})
        top["c4_cpu_value"] = cb.get("value"); top["c4_cpu_ratio"] = (c4["value"] / cb["value"]) if cb.get("value") else None
    c5 = subs.get("c5-mini")
    if isinstance(c5, dict) and "error" not in c5:
        top["c5_mini_tflops"] = c5.get("roofline", {}).get("achieved"); top["c5_mini_mfma_frac"] = c5.get("roofline", {}).get("frac")
    line.update({k: v for k, v in top.items() if v is not None})
    for k in ("roofline", "roofline_prefilter"):
        if k in out:
            line[k] = _slim_roof(out[k])
    if "roofline_end_to_end" in out:
        line["roofline_end_to_end"] = _pick(out["roofline_end_to_end"], ("bound", "achieved", "peak", "unit", "frac", "bytes_per_alignment"))
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _slim_cpu(out["cpu_baseline"])
    for k in ("stages_s_per_step", "prefilter_kernel_ms_per_step", "sw_kernel_ms_per_step", "rccl_ranks", "exchange_rank0_s_per_step", "exchange_bytes_per_step_all_ranks", "phases_max_over_ranks_s_per_step",
              "wall_s_per_step", "load_s_per_step", "value_disk_to_cluster_db", "encoder_residues_per_s"):
        if k in out:
            line[k] = out[k]
    if "rounds_last_step" in out:
        line.update(_pick(_slim_record(out), ("rounds", "rounds_columns")))
    if subs:
        line["configs"] = {k: _slim_record(v) for k, v in subs.items()}
        errs = {k: v["error"] for k, v in subs.items() if isinstance(v, dict) and "error" in v}
        if errs:
            line["errors"] = errs
    if "optional_rules" in out:
        line["optional_rules"] = out["optional_rules"]
    line["detail_file"] = detail

    def sig(x):      # six significant digits are more than any of these measurements carry; the full precision is in the detail file
        if isinstance(x, float):
            return float("%.6g" % x)
        if isinstance(x, dict):
            return {k: sig(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [sig(v) for v in x]
        return x
    return sig(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS) + ["c5"])
    ap.add_argument("--proteomes", type=int)
    ap.add_argument("--families", type=int)
    ap.add_argument("--len-scale", type=float)
    ap.add_argument("--options")
    ap.add_argument("--target-shards", type=int, default=int(os.environ.get("UC_TARGET_SHARDS", "0")),
                    help="T of the Q x T grid for N > 1 (0 = one target shard per GPU, the north-star layout)")
    ap.add_argument("--workdir", default=os.environ.get("UC_BENCH_DIR", "/tmp/uc_bench"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the disk-to-TSV and default-workflow legs")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-sub-records", action="store_true", help="skip the c3 / c5-mini / c4 sub-records and the one-shot-process leg of the default line")
    ap.add_argument("--no-c4", action="store_true", help="skip the nominal configs[3] sub-record (2000 proteomes: ~6 min incl. database generation and its CPU leg)")
    ap.add_argument("--optional-rules", action="store_true", help="add the timed leg of the optional rule UC-1/L (length gate, default off) to the default line")
    ap.add_argument("--detail-dir", help="where the full record of the run goes (default: gpurun_out/ of the repository, else --workdir)")
    ap.add_argument("--full-line", action="store_true", help="print the full record (every sub-record with its prose, ~20 kB) instead of the slim line + detail file")
    ap.add_argument("--workflow", choices=["plain", "default"], help="plain = the all-vs-all step (--single-step-clustering semantics; default for c2/c3/c4-lite); "
                    "default = what cluster.rs:45-49 triggers: pre-step + 3-step cascade through one uc_cluster call (default for c4)")
    args = ap.parse_args()
    for kv in filter(None, os.environ.get("UC_BENCH_PROTEOMES", "").split(",")):      # test hook, e.g. "c2=5,c3=5": the named configs at a toy size
        k, v = kv.split("=")
        CONFIGS[k] = (int(v),) + CONFIGS[k][1:]
    proteomes, families, len_scale, seed, options, label = CONFIGS[args.config] if args.config != "c5" else (5, 6000, 1.0, 0x5EED0005, "-c 0.8", "BASELINE configs[4]")
    custom = any(v is not None for v in (args.proteomes, args.families, args.len_scale, args.options))
    proteomes = args.proteomes if args.proteomes is not None else proteomes
    families = args.families if args.families is not None else families
    len_scale = args.len_scale if args.len_scale is not None else len_scale
    options = args.options if args.options is not None else options
    if args.config != "c5" and (args.workflow or ("default" if args.config == "c4" else "plain")) == "default":
        if args.gpus != 1:
            raise SystemExit("--workflow default is a single-process line (uc_cluster spreads over GPUs itself with '--gpus N' in --options)")
        print(json.dumps(slim_line(bench_workflow(args, proteomes, families, len_scale, seed, options, label, custom), args)))
        return

    import torch
    import unicore_amd as U

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the N > 1 code path = everything `multi` switches on: RCCL communicator, gloo control plane, max/sum over ranks, the configs[2] leg over all
    # ranks, the phase maxima.  UC_BENCH_FORCE_MULTI=1 takes it with ONE rank (a real ncclCommInitRank of world 1 under torch.distributed.run): the
    # single-GPU box proves the line's N > 1 keys before the 8-GPU node needs them (tests/test_multi_gpu.py)
    multi = world > 1 or os.environ.get("UC_BENCH_FORCE_MULTI") == "1"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one process per GPU, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    if args.config == "c5":
        return bench_c5(args, world, rank, local_rank, multi)
    comm = None
    dist = None
    rccl_ranks = 0
    if multi:
        import torch.distributed as dist
        dist.init_process_group("gloo")                     # control plane only: RCCL id, barriers, max of the timing
        uid = [U.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = U.Comm(uid[0], rank, world, device=local_rank)   # collective: ncclCommInitRank inside the library
        rccl_ranks = comm.info()[0]                          # what RCCL itself says (ncclCommCount)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    threads = max(1, (os.cpu_count() or 1) // world)

    def run_config(proteomes, families, len_scale, seed, options, label, custom, steps, warmup):
        """K timed passes of one workload; returns (the line as a dict on rank 0 / None elsewhere, DB prefix, n, workdir)"""
        workdir = os.path.join(args.workdir, "p%d_f%d_s%g_%x" % (proteomes, families, len_scale, seed))
        if rank == 0:
            gen_db(workdir, proteomes, families, len_scale, seed)
        barrier()
        prefix = os.path.join(workdir, "db")
        lens = read_lens(prefix)
        n = len(lens)
        eng = U.Engine(options, threads=threads, verbosity=1, device=local_rank)
        eng.load_db(prefix)                      # H2D upload: outside the timed region (inputs resident in HBM)
        assign = None
        for _ in range(warmup):
            assign, _ = eng.cluster_step(comm, args.target_shards)
        eng.reset_stats()
        barrier()
        t0 = time.perf_counter()
        n_aln = 0
        for _ in range(steps):
            assign, a = eng.cluster_step(comm, args.target_shards)
            n_aln += a
        barrier()
        dt = time.perf_counter() - t0
        st = eng.stats()
        eng.close()
        xbytes = 0
        if multi:
            v = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            dt = float(v.item())
            c = torch.tensor([n_aln, st["exchange_bytes"]], dtype=torch.int64)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            n_aln, xbytes = int(c[0].item()), int(c[1].item())
            ph = torch.tensor(st["phase_seconds"], dtype=torch.float64)
            dist.all_reduce(ph, op=dist.ReduceOp.MAX)
            st["phase_seconds"] = ph.tolist()
        if rank != 0:
            return None, prefix, n, workdir
        steps_ = max(steps, 1)
        sw_s = st["sw_kernel_ms"] / 1e3
        pre_s = st["prefilter_kernel_ms"] / 1e3
        achieved = st["sw_algorithmic_bytes"] / sw_s / 1e9 if sw_s > 0 else 0.0
        cells_alg = st["cells_fwd"] + st["cells_rev"] + st["cells_start"] + st["cells_tb"]     # what the spec asks for (oracle counts; cells_tb = traceback boxes, 0 without --min-seq-id)
        cells_run = st["cells_run"]                                            # what the kernels executed (mutual hits share a DP, flagged pairs run twice)
        ab = dict(zip(U.STAGES, st["algorithmic_bytes"]))
        pre_bytes = ab["index"] + ab["kmer"] + ab["ungapped"] + ab["select"]
        all_bytes = pre_bytes + ab["gapped"] + ab["setcover"]
        swt, pft = pmc_json("sw_traffic.json"), pmc_json("prefilter_traffic.json")
        is_c2 = not custom and label == CONFIGS["c2"][5]                       # the PMC files were collected on configs[1]
        n_streams = int(os.environ.get("UC_STREAMS", "8"))
        Q, T = (world // (args.target_shards or world), args.target_shards or world) if multi else (1, 1)
        out = {
            "metric": "3Di alignments/sec (cluster path)",
            "value": n_aln / dt, "unit": "alignments/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * dt / steps_, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": "%s: %d synthetic proteomes, %d seqs, %d residues, options '%s', plain all-vs-all step (--single-step-clustering "
                                   "semantics), gen_synth seed %#x, synthetic stand-in 3Di matrix" % (label if not custom else "custom size", proteomes, n, int(lens.sum()), options, seed),
                       "alignments_per_step": n_aln // steps_, "clusters": int((assign == np.arange(n)).sum()) if assign is not None else None,
                       "parallelism": ("%d query groups x %d target shards; exchange 1: shard lists to the query's home rank (ragged all-to-all, grouped ncclSend/ncclRecv "
                                       "from the C library), merge + top-M of 1/N of the queries per rank; exchange 2: surviving pairs to their owner rank; "
                                       "edges to rank 0" % (Q, T)) if multi else "single GPU"},
            "rccl_ranks": rccl_ranks,
            "value_definition": "DB resident in HBM at the start of the timed region; see value_disk_to_tsv for SURVEY.md 8(d)'s disk -> clust.tsv wall",
            # dominant kernel: the gapped SW (all classes and passes)
            "roofline": {"bound": "hbm", "kernel": "sw_pk_kernel + sw_group_kernel (gapped 3Di+AA SW, all classes and passes)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": ((swt["fetch_size_kib"] + swt["write_size_kib"]) * 1024.0 / swt["sw_launches"]) if (swt and is_c2) else None,
                         "algorithmic_bytes_per_launch": st["sw_algorithmic_bytes"] / max(st["sw_kernel_launches"], 1),
                         "algorithmic_bytes_per_step": st["sw_algorithmic_bytes"] / steps_, "kernel_ms_per_step": st["sw_kernel_ms"] / steps_,
                         "avg_launch_ms": st["sw_kernel_ms"] / max(st["sw_kernel_launches"], 1),
                         "launches": st["sw_kernel_launches"], "streams": n_streams,
                         "launch_overlap": ("the length classes of a pass run concurrently on %d HIP streams; avg_launch_ms = HIP-event time of the fork/join regions on the engine "
                                            "stream / launches.  With UC_STREAMS=1 the launches are serialized and the per-kernel durations of a rocprofv3 trace add up to that event "
                                            "time (profiles/round2/r3d_serial_summary.txt)" % n_streams),
                         "note": "integer-VALU-bound by design (SURVEY.md 8d): the meaningful fraction is valu_frac = cells_run x valu_ops_per_cell / (kernel s x valu_peak_lane_ops)",
                         "cells_run_per_step": cells_run / steps_, "cells_algorithmic_per_step": cells_alg / steps_,
                         "valu_gcups": cells_run / sw_s / 1e9 if sw_s > 0 else 0.0,
                         "valu_gcups_algorithmic": cells_alg / sw_s / 1e9 if sw_s > 0 else 0.0,
                         "valu_peak_lane_ops": VALU_PEAK_LANE_OPS, "valu_ops_per_cell": VALU_OPS_PER_CELL,
                         "valu_frac": (cells_run * VALU_OPS_PER_CELL / sw_s / VALU_PEAK_LANE_OPS) if sw_s > 0 else 0.0,
                         # MI355X_MICROARCH.md words the SIMDs as 32 lanes / clock (78.6 T lane-ops/s); the measured issue rate of the int / packed VALU
                         # instructions this kernel uses is 16 lanes / clock (tools/ubench/valu_rate.hip): both fractions, so neither can be misread
                         "valu_peak_lane_ops_guide_nominal": 2 * VALU_PEAK_LANE_OPS,
                         "valu_frac_of_guide_nominal_peak": (cells_run * VALU_OPS_PER_CELL / sw_s / (2 * VALU_PEAK_LANE_OPS)) if sw_s > 0 else 0.0},
            # the HBM-bound stages E1-E4 (index, similar-k-mer match + double-hit filter, ungapped, top-M)
            "roofline_prefilter": {"bound": "hbm", "kernels": "kmer_extract, sim_runs, filter, compact, diag_select, ungapped, select/rank/scatter + rocPRIM sorts",
                                   "algorithmic_bytes_per_step": pre_bytes / steps_, "kernel_ms_per_step": 1e3 * pre_s / steps_,
                                   "achieved": pre_bytes / pre_s / 1e9 if pre_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": (pre_bytes / pre_s / 1e9 / HBM_PEAK_GBS) if pre_s > 0 else 0.0,
                                   "frac_of_measured_copy_bw": (pre_bytes / pre_s / 1e9 / HBM_COPY_GBS) if pre_s > 0 else 0.0,
                                   "formula": "8 B x n_sim_kmers + 6 B x n_kmer_hits + 8 B x n_candidates (kmer) + 6 B x n_index_entries + 8 B x 20^6 (index) + ungapped + 16 B x "
                                              "n_candidates (select), all from counts_rank0_per_step / algorithmic_bytes_per_step of this line, divided by kernel_ms_per_step",
                                   "traffic_per_step": pft["bytes_per_step"] if (pft and is_c2) else None,
                                   "traffic_source": pft["source"] if (pft and is_c2) else None},
            "roofline_end_to_end": {"bound": "hbm", "algorithmic_bytes_per_step": all_bytes / steps_,
                                    "bytes_per_alignment": all_bytes / max(n_aln, 1) if not multi else None,
                                    "achieved": all_bytes / dt / 1e9, "peak": HBM_COPY_GBS, "unit": "GB/s", "frac": all_bytes / dt / 1e9 / HBM_COPY_GBS,
                                    "note": "SURVEY.md 8(d): sum of per-stage algorithmic bytes / (wall x 6.29 TB/s); rank 0's bytes when N > 1"},
            "algorithmic_bytes_per_step": {k: v / steps_ for k, v in ab.items()},
            "stages_s_per_step": {k: v / steps_ for k, v in zip(U.STAGES, st["stage_seconds"])},
            "prefilter_kernel_ms_per_step": st["prefilter_kernel_ms"] / steps_,
            "sw_kernel_ms_per_step": st["sw_kernel_ms"] / steps_,
            "sw_dp_runs_per_step": st["n_sw_runs"] // steps_,
            "exchange_rank0_s_per_step": st["exchange_seconds"] / steps_, "exchange_bytes_per_step_all_ranks": xbytes // steps_,
            "counts_rank0_per_step": {k: st[k] // steps_ for k in ("n_index_entries", "n_sim_kmers", "n_kmer_hits", "n_filtered_hits", "n_candidates", "n_prefilter_hits",
                                                                   "n_gapped_alignments", "n_start_alignments", "n_pk_reruns")},
            "edges_last_step": st["n_edges"],
        }
        if multi:
            out["phases_max_over_ranks_s_per_step"] = {k: v / steps_ for k, v in zip(U.PHASES, st["phase_seconds"])}
        return out, prefix, n, workdir

    out, prefix, n, workdir = run_config(proteomes, families, len_scale, seed, options, label, custom, args.steps, args.warmup)

    if rank == 0:
        if not multi and not args.no_extra_legs:
            # uc_cluster + uc_createtsv, DB on disk (page cache warm) -> clust.tsv
            outp = os.path.join(workdir, "bench_clust")
            walls = []
            for _ in range(3):
                t1 = time.perf_counter()
                s1 = U.cluster(prefix, outp + "_cluster", os.path.join(workdir, "tmp"), options + " --single-step-clustering", threads=threads)
                U.createtsv(prefix, outp + "_cluster", outp + ".tsv")
                walls.append(time.perf_counter() - t1)
            tsv_ok = s1["n_clusters"] == out["config"]["clusters"] and sum(1 for _ in open(outp + ".tsv")) == n
            out["value_disk_to_tsv"] = {"value": s1["n_gapped_alignments"] / min(walls), "unit": "alignments/s", "wall_s_best": min(walls),
                                        "wall_s_mean": sum(walls) / len(walls), "value_mean": s1["n_gapped_alignments"] / (sum(walls) / len(walls)),
                                        "alignments": s1["n_gapped_alignments"], "clusters": s1["n_clusters"], "tsv_rows": n, "same_clusters_as_steps": bool(tsv_ok),
                                        "what": "uc_cluster('%s --single-step-clustering') + uc_createtsv from the DB files to clust.tsv (SURVEY.md 8d), 3 runs" % options}
            walls = []
            for _ in range(2):
                t1 = time.perf_counter()
                s2 = U.cluster(prefix, outp + "_cluster", os.path.join(workdir, "tmp"), options, threads=threads)
                U.createtsv(prefix, outp + "_cluster", outp + ".tsv")
                walls.append(time.perf_counter() - t1)
            out["workflow_default"] = {"what": "uc_cluster('%s') + uc_createtsv: Foldseek's default workflow (linear-time pre-step + 3-step cascade), disk -> clust.tsv" % options,
                                       "wall_s_best": min(walls), "alignments": s2["n_gapped_alignments"], "clusters": s2["n_clusters"],
                                       "value": s2["n_gapped_alignments"] / min(walls), "unit": "alignments/s"}
            U.rmdb(outp + "_cluster")
            if not args.no_sub_records:
                U.lib().uc_release_scratch()     # this process's parked work buffers (tens of GB) go back first: the spawned processes allocate their own
                out["value_one_shot_processes"] = one_shot_processes(prefix, workdir, options, s1["n_gapped_alignments"], s2["n_gapped_alignments"])
        if not multi and not args.no_cpu_baseline:
            cb = cpu_baseline(prefix, options, n, args.cpu_seconds)
            best = max(cb, key=lambda d: d["value"])
            out["cpu_baseline"] = best                       # the faster CPU leg is THE baseline ...
            out["cpu_baselines"] = cb                        # ... both are reported
    if not multi and args.config == "c2" and not custom and not args.no_sub_records and not args.no_extra_legs:
        # every sub-record is guarded on its own: an exception, an out-of-memory or a missing tool in one leg leaves {"error": ...} in its
        # place and the headline, the other records and the line itself stand
        subs = {}

        def guarded(name, fn):
            try:
                U.lib().uc_release_scratch()
                subs[name] = fn()
            except BaseException as e:          # noqa: BLE001 (SystemExit of a nested leg included: the line must still be printed)
                if isinstance(e, KeyboardInterrupt):
                    raise
                subs[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

        def leg_c3():
            # north_star's quoted 1-GPU target size (BASELINE configs[2] on one GPU): ONE timed pass after one untimed pass (a cold pass spends
            # ~6 s of its 46 s in first-touch hipMalloc of ~150 GB of work buffers), its own roofline blocks and CPU sample
            p3 = CONFIGS["c3"]
            o3, prefix3, n3, _ = run_config(p3[0], p3[1], p3[2], p3[3], p3[4], p3[5], False, 1, 1)
            for k in ("metric", "unit", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "value_definition", "rccl_ranks"):
                o3.pop(k, None)
            if not args.no_cpu_baseline:
                cb3 = cpu_baseline(prefix3, p3[4], n3, min(args.cpu_seconds, 12.0))
                o3["cpu_baseline"] = max(cb3, key=lambda d: d["value"])
            # ... and the same database through the DEFAULT workflow (what cluster.rs:45-49 triggers), disk -> clust.tsv, second of two calls
            U.lib().uc_release_scratch()
            outp3 = os.path.join(os.path.dirname(prefix3), "bench_clust")
            for _ in range(2):
                t1 = time.perf_counter()
                s3 = U.cluster(prefix3, outp3 + "_cluster", os.path.join(os.path.dirname(prefix3), "tmp"), p3[4], threads=threads)
                U.createtsv(prefix3, outp3 + "_cluster", outp3 + ".tsv")
                w3 = time.perf_counter() - t1
            U.rmdb(outp3 + "_cluster")
            o3["workflow_default"] = {"what": "uc_cluster('%s') + uc_createtsv: pre-step + 3-step cascade, disk -> clust.tsv, warm call" % p3[4], "wall_s": w3,
                                      "alignments": s3["n_gapped_alignments"], "clusters": s3["n_clusters"], "value": s3["n_gapped_alignments"] / w3, "unit": "alignments/s",
                                      "sw_kernel_ms": s3["sw_kernel_ms"], "prefilter_kernel_ms": s3["prefilter_kernel_ms"]}
            return o3

        def leg_c4():
            # BASELINE configs[3] at its NOMINAL size (2000 proteomes, 6.36 M sequences, "-c 0.8 --min-seq-id 0.3 -s 7.5"): ONE uc_cluster call through the
            # default workflow (what cluster.rs:35,45-49 forwards), from the DB files; its own rooflines, per-round records and the round-by-round CPU leg
            p4 = CONFIGS["c4"]
            o4 = bench_workflow(args, p4[0], p4[1], p4[2], p4[3], p4[4], p4[5], False, steps=1, warmup=0, cpu_seconds=min(args.cpu_seconds, 10.0))
            for k in ("metric", "unit", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "value_definition", "n_gpus"):
                o4.pop(k, None)
            return o4

        guarded("c3", leg_c3)
        guarded("c5-mini", lambda: c5_mini(args))
        if not args.no_c4:
            guarded("c4", leg_c4)
        out["configs"] = subs
    if not multi and args.config == "c2" and not custom and not args.no_extra_legs and args.optional_rules:
        # optional rule UC-1/L (default OFF; opt-in leg since r06: --optional-rules): the headline workload with MMseqs2's length gate in
        # front of the gapped stage - what the step costs if Foldseek's aligner skips the pairs whose lengths alone rule the coverage threshold out (believed,
        # EXT-UNVERIFIED).  value counts the pairs that WERE aligned, not the listed ones.
        U.lib().uc_release_scratch()
        p2 = CONFIGS["c2"]
        og, _, _, _ = run_config(p2[0], p2[1], p2[2], p2[3], p2[4] + " --length-gate 1", p2[5], True, max(1, min(args.steps, 3)), 1)
        out["optional_rules"] = {"UC-1/L": {"options": p2[4] + " --length-gate 1", "default": "off", "ms_per_step": og["ms_per_step"], "value": og["value"], "unit": "alignments/s",
                                            "alignments_per_step": og["config"]["alignments_per_step"], "listed_pairs_per_step": out["config"]["alignments_per_step"],
                                            "clusters": og["config"]["clusters"], "sw_kernel_ms_per_step": og["roofline"]["kernel_ms_per_step"],
                                            "cells_run_per_step": og["roofline"]["cells_run_per_step"]}}
    if multi and args.config == "c2" and not custom and not args.no_sub_records:
        # the configuration BASELINE names for the 8-GPU node (configs[2]: 500 proteomes, target DB sharded across the ranks): ONE timed pass
        # after one warm-up pass with all N ranks; the headline above stays configs[1] so that the N = 1 point of a scaling run agrees with BENCH
        U.lib().uc_release_scratch()
        p3 = CONFIGS["c3"]
        o3, _, _, _ = run_config(p3[0], p3[1], p3[2], p3[3], p3[4], p3[5], False, 1, 1)
        if rank == 0:
            for k in ("metric", "unit", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "value_definition"):
                o3.pop(k, None)
            out["configs"] = {"c3": o3}
    if rank == 0:
        print(json.dumps(slim_line(out, args)))
    if multi:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
