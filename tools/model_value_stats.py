import os, sys, tempfile
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import _gmg_pkg, models64
gmg = _gmg_pkg.load(); gmg.init(0)
def stats(m, name):
    mip, prob = m.tables()
    W, D, P, N = m.params
    p = prob.reshape(-1)
    used = p[np.repeat(mip.reshape(-1) > -2, 4)] if mip.size * 4 == p.size else p
    nz = used[(used != 0) & np.isfinite(used)]
    e = np.frexp(np.abs(nz))[1]
    print("%-28s nodes used %7d  values<-1e30: %6d  exp range %d..%d  min %.3g max %.3g" % (name, (mip > -2).sum(), (used < -1e30).sum(), e.min(), e.max(), nz.min(), nz.max()))
t = tempfile.mkdtemp()
for m, p in models64.gene_models(gmg, t, 8): stats(m, os.path.basename(p))
for m, p in models64.period1_models(gmg, t, 9): stats(m, os.path.basename(p))
g = models64.genome()
big = gmg.Icm.train([g[i:i + 900] for i in range(0, 600_000, 900)], 12, 7, 3); stats(big, "trained on 600 kb")
big = gmg.Icm.train([g[i:i + 1000] for i in range(0, 1_600_000, 1000)], 12, 7, 1); stats(big, "p1 trained on 1.6 Mb")
