import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import unicore_amd as U
from oracle import prostt5_ref as R
path = "/tmp/t5_hist.gguf"
R.write_synthetic_gguf(path, R.default_config(n_layers=24), seed=0x5EED0005)
rng = np.random.default_rng(1)
fam = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), 300))
mut = list(fam)
for p in rng.choice(300, 40, replace=False): mut[p] = rng.choice(list("ACDEFGHIKLMNPQRSTVWY"))
other = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), 300))
enc = U.T5Encoder(path)
c = enc.encode([fam, "".join(mut), other])
print("state histogram:", sorted(collections.Counter(c[0].tolist()).items()))
print("identity fam vs mutant (12%% AA subst): %.2f   fam vs unrelated: %.2f" % ((c[0] == c[1]).mean(), (c[0] == c[2]).mean()))
