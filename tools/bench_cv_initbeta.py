"""cv_iht with init_beta=true (the setting of the reference's large real runs, manuscript/UKBB_hyptertension/ukbb.jl:16-18) at
n = 500k, p = 1M, Normal, path = 1:20, 5 folds; MIH_FITS limits the grid (path prefix) for a quick look."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
n, p = int(os.environ.get("MIH_N", 500_000)), int(os.environ.get("MIH_P", 1_000_000))
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
y = x.xv_sparse(supp, rng.standard_normal(10) * 0.5) + 1.0 + rng.standard_normal(n)
folds = m.hash_folds(n, 5)
npath = int(os.environ.get("MIH_NPATH", 20))
for ib in (False, True, True):
    m.profile_read(x, reset=True); m.profile_enable(x, True)
    t0 = time.perf_counter()
    mse, raw = m.cv_iht(y, x, None, path=range(1, npath + 1), q=5, folds=folds, verbose=False, return_raw=True, init_beta=ib)
    dt = time.perf_counter() - t0
    m.profile_enable(x, False)
    ms, launches = m.profile_read(x, reset=True)
    print(f"init_beta={ib}: {dt:.3f} s for {5 * npath} fits, {launches} X'r launches ({ms:.0f} ms), best k {int(np.argmin(mse)) + 1}, checksum {raw.sum()!r}", flush=True)
