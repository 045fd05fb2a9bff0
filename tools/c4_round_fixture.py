#!/usr/bin/env python3
"""tools/c4_round_fixture.py dump|oracle [--name c4] — the committed per-round oracle records of an at-size default-workflow test.

BASELINE configs[3] at its nominal 2000 proteomes is out of the CPU oracle's reach end to end (~14 h on the build container's 8 cores), and the
oracle's per-round passes over four sub-databases of up to 1.9 G residues cost the GPU box's host more minutes than the HIP pass itself.  So the
oracle side of `tests/test_workflow_gpu.py::test_default_workflow_at_size[c4]` is computed ONCE, in the build container, in two steps:

  dump    (GPU box, through gpurun)   one uc_cluster call with the workflow observer: the sequence set of every round (a packed bitmap per round),
                                      its k-mer threshold and pair count, the sha256 of the resulting clust.tsv -> gpurun_out/<name>_rounds_dump.npz
  oracle  (build container, no GPU)   regenerates the same database, re-derives the test's query sample of every round (same generator, same seed)
                                      and runs the CPU oracle on every round's sub-database -> tests/golden/<name>_rounds.npz

The test then asserts (i) that its run goes through the SAME round sets (sha256 of the id arrays) and (ii) every sampled record against the
committed oracle records.  What the fixture does NOT claim: that the round sets themselves are the oracle's (they come from the HIP run; each is
the set cover of the round before, which the sampled records of that round pin only statistically) - the whole-file goldens at 500 proteomes
(tests/golden/c4-500_workflow_sha.json, c3_workflow_sha.json) carry that statement."""
import argparse, hashlib, json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["dump", "oracle"])
    ap.add_argument("--name", default="c4")
    ap.add_argument("--work", default="/tmp/uc_round_fixture")
    ap.add_argument("--dump", default=None, help="the dump file (default gpurun_out/<name>_rounds_dump.npz)")
    a = ap.parse_args()
    import util
    import test_workflow_gpu as W
    cfg = W.AT_SIZE[a.name]
    dump = a.dump or os.path.join(ROOT, "gpurun_out", "%s_rounds_dump.npz" % a.name)
    os.makedirs(a.work, exist_ok=True)
    t0 = time.time()
    db = util.gen_synth_db(os.path.join(a.work, "db_%s" % a.name), cfg["proteomes"], cfg["seed"], 6000, 1.0)
    print("[%6.0f s] database %s" % (time.time() - t0, db), file=sys.stderr, flush=True)
    if a.mode == "dump":
        import torch  # noqa: F401
        import unicore_amd as U
        rounds = []

        def hook(rnd, ids, kthr, view):
            rounds.append((rnd, ids, kthr, view.hits_size()))
        U.set_round_hook(hook)
        out = os.path.join(a.work, "clust_%s" % a.name)
        try:
            st = U.cluster(db, out + "_cluster", os.path.join(a.work, "tmp"), cfg["opts"], threads=16)
        finally:
            U.set_round_hook(None)
        U.createtsv(db, out + "_cluster", out + ".tsv")
        n = st["n_seqs"]
        z = {"sequences": n, "n_rounds": len(rounds), "hip_tsv_sha256": hashlib.sha256(open(out + ".tsv", "rb").read()).hexdigest(),
             "hip_clusters": st["n_clusters"], "hip_alignments": st["n_gapped_alignments"], "options": cfg["opts"]}
        for k, (rnd, ids, kthr, pairs) in enumerate(rounds):
            bm = np.zeros(n, np.uint8); bm[ids] = 1
            z["r%d_round" % k] = rnd; z["r%d_bitmap" % k] = np.packbits(bm); z["r%d_kmer_thr" % k] = kthr; z["r%d_n_pairs" % k] = pairs
        os.makedirs(os.path.dirname(dump), exist_ok=True)
        np.savez_compressed(dump, **z)
        print(json.dumps({k: (v if not isinstance(v, np.ndarray) else int(v.size)) for k, v in z.items()}), flush=True)
        print("[%6.0f s] wrote %s" % (time.time() - t0, dump), file=sys.stderr)
        return
    from oracle import oracle_py as O
    z = np.load(dump)
    odb = O.OracleDb(db)
    n = int(z["sequences"])
    assert odb.n == n and str(z["options"]) == cfg["opts"]
    f = {"sequences": n, "per_round": cfg["per_round"], "sample_seed": W.SAMPLE_SEED, "options": cfg["opts"], "hip_tsv_sha256": str(z["hip_tsv_sha256"]),
         "hip_clusters": int(z["hip_clusters"]), "round_sizes": []}
    for k in range(int(z["n_rounds"])):
        rnd = int(z["r%d_round" % k])
        ids = np.flatnonzero(np.unpackbits(z["r%d_bitmap" % k])[:n]).astype(np.uint32)
        qs = W.sample_queries(len(ids), rnd, cfg["per_round"], W.SAMPLE_SEED)
        t1 = time.time()
        ref = W.oracle_round(O, odb, cfg["opts"], rnd, ids, int(z["r%d_kmer_thr" % k]), qs)
        if rnd < 0:
            assert int(ref["n_pairs"]) == int(z["r%d_n_pairs" % k]), "the pre-step's pair count differs from the HIP run's"
        f["round_sizes"].append(len(ids))
        f["r%d_ids_sha256" % k] = hashlib.sha256(ids.tobytes()).hexdigest()
        f["r%d_kmer_thr" % k] = int(z["r%d_kmer_thr" % k])
        f["r%d_queries" % k] = np.asarray(qs, np.uint32)
        for key, v in ref.items():
            v = np.asarray(v)
            if v.ndim and v.dtype.kind in "iu" and v.size and key != "hit_t" and np.abs(v.astype(np.int64)).max() < 32768:
                v = v.astype(np.int16)
            f["r%d_%s" % (k, key)] = v
        print("[%6.0f s] round %d: %d sequences, %d queries, %d records (%.0f s)" % (time.time() - t0, rnd, len(ids), len(qs), int(ref["cnt"].sum()), time.time() - t1),
              file=sys.stderr, flush=True)
    f["round_sizes"] = np.array(f["round_sizes"], np.int64)
    out = os.path.join(ROOT, "tests", "golden", "%s_rounds.npz" % a.name)
    np.savez_compressed(out, **f)
    print("wrote %s (%d bytes)" % (out, os.path.getsize(out)))


if __name__ == "__main__":
    main()
