#!/usr/bin/env python3
"""tools/createdb_replicas_at_size.py [--proteomes 20] [--replicas 3] — `uc_createdb` with the full-geometry 24-block synthetic ProstT5 model on a synthetic proteome
set, once with ONE encoder replica and once with N (on the single-GPU box: N replicas sharing the device, UC_VIRTUAL_GPUS=1; on a node: num_gpus = N real devices):
every file of the two databases must be byte-identical; prints the stats of both runs (tokens per replica, GPU time of the slowest replica, wall) as one JSON line.
The at-size companion of tests/test_t5.py::test_createdb_on_n_encoder_replicas_writes_the_same_database (tiny model there)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import torch  # noqa: F401
import bench
import unicore_amd as U
import make_t5_full_depth as F

ap = argparse.ArgumentParser()
ap.add_argument("--proteomes", type=int, default=20)
ap.add_argument("--replicas", type=int, default=3)
ap.add_argument("--work", default="/tmp/uc_bench")
a = ap.parse_args()
seed = 0x5EED0005
db = bench.gen_db(os.path.join(a.work, "p%d_f6000_s1_%x" % (a.proteomes, seed)), a.proteomes, 6000, 1.0, seed)
aa = [e.decode() for e in open(db, "rb").read().split(b"\n\0")[:-1]]
names = [l.split("\t")[1] for l in open(db + ".lookup")]
fa = os.path.join(a.work, "createdb_replicas_%d.fasta" % a.proteomes)
with open(fa, "w") as f:
    for n, s in zip(names, aa):
        f.write(">%s\n%s\n" % (n, s))
gguf = F.ensure_gguf()
ndev = torch.cuda.device_count()
if a.replicas > ndev:
    os.environ["UC_VIRTUAL_GPUS"] = "1"
out = {"proteomes": a.proteomes, "sequences": len(aa), "residues": sum(len(x) for x in aa), "visible_gpus": ndev, "runs": []}
FILES = ("", "_ss", "_h", ".index", "_ss.index", "_h.index", ".dbtype", "_ss.dbtype", "_h.dbtype", ".lookup", "_ss.source")
for n in (1, a.replicas):
    o = os.path.join(a.work, "createdb_replicas_out_%d" % n)
    t = time.perf_counter()
    st = U.createdb(fa, o, gguf, num_gpus=n)
    st["wall_s"] = time.perf_counter() - t
    st["tflops_all_replicas"] = st["flops"] / (st["gpu_ms"] * 1e-3) / 1e12 if st["gpu_ms"] else None
    out["runs"].append(st)
a1, aN = os.path.join(a.work, "createdb_replicas_out_1"), os.path.join(a.work, "createdb_replicas_out_%d" % a.replicas)
out["files_identical"] = all(open(a1 + s, "rb").read() == open(aN + s, "rb").read() for s in FILES)
assert out["files_identical"], "the databases of 1 and %d replicas differ" % a.replicas
print(json.dumps(out))
