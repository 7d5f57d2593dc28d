"""Cost of debias=true at BASELINE configs[2] size (n=500k, p=1M, k=200, Normal): ms per IHT iteration."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
n, p, k = 500_000, int(os.environ.get("MIH_P", 1_000_000)), 200
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, size=k, replace=False))
y = x.xv_sparse(supp, rng.standard_normal(k)) + 1.0 + rng.standard_normal(n)
for db in (False, True):
    res = m.fit_iht(y, x, None, k=k, debias=db, verbose=False, max_iter=30)
    print(f"debias={db}: {res.iter} iterations, {1e3 * res.time / res.iter:.1f} ms/iteration, logl {res.logl:.4f}, "
          f"recovered {np.intersect1d(np.flatnonzero(res.beta), supp).size}/{k}", flush=True)
