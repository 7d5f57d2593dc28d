"""Multivariate cross-validation at BASELINE configs[4] size (r = 10 traits, n = 500k, p = 1M): 3 folds x path = [100, 300, 500]."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mendeliht_amd as m
hash_folds = m.hash_folds
n, p, r = 500_000, int(os.environ.get("MIH_P", 1_000_000)), 10
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(3)
lin = rng.choice(r * p, 300, replace=False)
Y = rng.standard_normal((r, n))
for t in range(r):
    cols = np.unique(lin[lin % r == t] // r)
    Y[t] += x.xv_sparse(cols, rng.standard_normal(cols.size) * 0.3) + 1.0
folds = hash_folds(n, 3)
for rep in range(2):
    t0 = time.perf_counter()
    mse, raw = m.cv_iht(Y, x, None, path=[100, 300, 500], q=3, folds=folds, verbose=False, return_raw=True, max_iter=8)
    print(f"rep {rep}: 9 multivariate fits in {time.perf_counter() - t0:.2f} s, best k {[100, 300, 500][int(np.argmin(mse))]}, checksum {raw.sum():.10e}", flush=True)
