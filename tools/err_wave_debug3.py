import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gmg_pkg
gmg = _gmg_pkg.load(); gmg.init(0)
DATA = os.path.join(ROOT, "tests", "golden", "data")
nc = gmg.Icm.open(os.path.join(DATA, "NC_000915.icm"))
rng = np.random.default_rng(99)
lengths = [0, 1, 5, 14, 15, 16, 17, 18, 33, 74, 75, 76, 99, 150, 231, 300, 301, 302, 400, 523, 700]
seqs = ["".join("acgt"[c] for c in rng.integers(0, 4, size=n)) for n in lengths]
reads = gmg.Reads.from_strings([seqs[18]])
gmg.set_option("mg_err_wave", 1); gmg.set_option("mg_err_tile", 0)
r = gmg.mg_score_reads(nc, gmg.Icm.indep(0.5), reads, allow_truncated=False, min_gene_len=60, allow_indels=True)
o = r[0]
for i in range(len(o)):
    print(i, o[i]["frame"], o[i]["stop_position"], o[i]["lo"], o[i]["hi"], o[i]["n_starts"])
