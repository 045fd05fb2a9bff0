import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
import torch
import unicore_amd as U
import subprocess
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for prot in (int(x) for x in sys.argv[1:]):
    d = "/tmp/uc_probe/p%d" % prot
    os.makedirs(d, exist_ok=True)
    if not os.path.exists(d + "/db.map"):
        subprocess.check_call([root + "/bin/gen_synth", d + "/db", str(prot), hex(0x5EED0004), "6000", "1.0"], stderr=subprocess.DEVNULL)
    e = U.Engine("-c 0.8 --min-seq-id 0.3 -s 7.5", threads=16, verbosity=3)
    e.load_db(d + "/db")
    t0 = time.time(); e.prefilter(); t1 = time.time(); e.align(); t2 = time.time()
    st = e.stats()
    print(json.dumps({"proteomes": prot, "n": e.n, "prefilter_s": t1 - t0, "align_s": t2 - t1, "hits": st["n_kmer_hits"], "sim": st["n_sim_kmers"], "filtered": st["n_filtered_hits"],
                      "aln": st["n_gapped_alignments"], "sw_ms": st["sw_kernel_ms"], "pre_ms": st["prefilter_kernel_ms"]}), flush=True)
    e.close()
