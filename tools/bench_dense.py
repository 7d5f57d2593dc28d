"""BASELINE configs[1]: fit_iht on a dense Float64 matrix 50000 x 100000, k=100, Normal (40 GB in HBM)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
n, p, k = 50_000, 100_000, 100
x = m.DenseMatrix.synthetic(n, p, seed=7)
B = x.algorithmic_bytes()
ms, cs = x.bench_xtv(-1, iters=5, warmup=1)
print(f"dense X'r {n}x{p}: {ms:.3f} ms  {B / ms / 1e6:.0f} GB/s ({B / ms / 1e6 / 80:.1f}% of 8 TB/s)")
rng = np.random.default_rng(1)
supp = np.sort(rng.choice(p, k, replace=False)); beta = rng.standard_normal(k)
y = x.xv_sparse(supp, beta) + 1 + rng.standard_normal(n)
t0 = time.perf_counter()
res = m.fit_iht(y, x, None, k=k, verbose=False)
dt = time.perf_counter() - t0
rec = np.intersect1d(np.flatnonzero(res.beta), supp).size
print(f"fit_iht dense: {res.iter} iterations in {res.time:.3f} s ({res.time / res.iter * 1e3:.2f} ms/iter; wall {dt:.2f} s), recovered {rec}/{k}, logl {res.logl:.3f}")
