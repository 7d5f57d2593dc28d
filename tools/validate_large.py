"""One-off parity check at scale (too slow for the test suite): fit_iht on n = 500 000 x p columns, k = 200,
Normal/Identity -- the bench.py workload with fewer columns -- on the GPU and with the CPU oracle on the same matrix."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
from oracle import oracle as O

try:
    lim = open("/sys/fs/cgroup/memory.max").read().strip()
except OSError:
    lim = "max"
print("cgroup memory.max:", lim, flush=True)
n, k = 500_000, 200
p = int(os.environ.get("MIH_P", 60_000))
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, size=k, replace=False))
beta = rng.standard_normal(k)
y = x.xv_sparse(supp, beta) + 1.0 + rng.standard_normal(n)
t0 = time.perf_counter(); res = m.fit_iht(y, x, None, k=k, verbose=False); tg = time.perf_counter() - t0
cols = x.export_bed()
ox = O.Mat.from_bed_columns(cols, n)
del cols
O.set_threads(int(os.environ.get("OMP_NUM_THREADS", 32)))
t0 = time.perf_counter(); o = O.fit_iht(ox, y, None, k=k); tc = time.perf_counter() - t0
same = np.array_equal(np.flatnonzero(res.beta), np.flatnonzero(o["beta"]))
nz = res.beta != 0
print(f"n={n} p={p} k={k}: GPU {res.iter} iterations in {tg:.2f} s, oracle {o['iter']} iterations in {tc:.1f} s; "
      f"same support: {same}; max |beta - beta_oracle| = {np.max(np.abs(res.beta - o['beta'])):.3e} "
      f"(max |beta| {np.abs(res.beta[nz]).max():.3f}); logl {res.logl!r} vs {o['logl']!r}; "
      f"true effects recovered {np.intersect1d(np.flatnonzero(res.beta), supp).size}/{k}")

if os.environ.get("MIH_CV"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    hash_folds = m.hash_folds
    eta = x.xv_sparse(supp[:10], beta[:10] * 0.5)
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    folds = hash_folds(n, 3)
    path = [5, 10, 15]
    t0 = time.perf_counter()
    mse = m.cv_iht(yb, x, None, d=m.Bernoulli(), l=m.LogitLink(), path=path, q=3, folds=folds, verbose=False)
    tg = time.perf_counter() - t0
    t0 = time.perf_counter()
    omse, _ = O.cv_iht(ox, yb, None, path=path, q=3, folds=folds, dist="bernoulli", link="logit")
    tc = time.perf_counter() - t0
    print(f"cv_iht Bernoulli/Logit path={path} q=3: GPU {tg:.2f} s, oracle {tc:.1f} s; losses {mse} vs {omse}; "
          f"max relative difference {np.max(np.abs(mse - omse) / omse):.3e}")
