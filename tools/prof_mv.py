"""One multivariate fit at BASELINE configs[4] size (r=10, k=500, n=500k, p=1M) for rocprofv3."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
n, p, r, k = 500_000, int(os.environ.get("MIH_P", 1_000_000)), 10, 500
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(3)
lin = rng.choice(r * p, k, replace=False)
Y = rng.standard_normal((r, n))
for t in range(r):
    cols = np.unique(lin[lin % r == t] // r)
    Y[t] += x.xv_sparse(cols, rng.standard_normal(cols.size) * 0.3) + 1.0
t0 = time.perf_counter()
res = m.fit_iht(Y, x, None, k=k, verbose=False, max_iter=8)
print(f"{res.iter} iterations, {1e3 * res.time / res.iter:.1f} ms/iteration, wall {time.perf_counter() - t0:.2f} s")
