import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
out = {}
for (n, p, r, k) in ((3001, 800, 10, 40), (20000, 1500, 7, 100), (5003, 400, 3, 9), (40000, 600, 12, 60)):
    x = m.SnpLinAlg.synthetic(n, p, seed=5)
    rng = np.random.default_rng(n)
    Y = rng.standard_normal((r, n))
    for t in range(r):
        cols = np.sort(rng.choice(p, 4, replace=False))
        Y[t] += x.xv_sparse(cols, rng.standard_normal(4))
    res = m.fit_iht(Y, x, None, k=k, verbose=False, max_iter=12)
    out[f"b{n}"] = res.beta; out[f"l{n}"] = np.array([res.logl, res.iter])
np.savez(sys.argv[1], **out)
