"""BASELINE configs[3] on one GPU (all 100 fits) and one rank's share of 8 (13 fits), twice each; MENDELIHT_XTV_MAX_OPS / MENDELIHT_CV_LANES select
the pass width and the number of lock-step lanes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
hash_folds = m.hash_folds
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = hash_folds(n, 5)
ref = None
for rep in range(2):
    for world in (1, 8):
        m.profile_read(x, reset=True); m.profile_enable(x, True)
        t0 = time.perf_counter()
        mse, raw = m.cv_iht(yb, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, rank=0, world=world,
                            d=m.Bernoulli(), l=m.LogitLink())
        dt = time.perf_counter() - t0
        m.profile_enable(x, False)
        ms, launches = m.profile_read(x, reset=True)
        if world == 1:
            if ref is None: ref = raw.copy()
            assert np.array_equal(raw, ref)
        print(f"max_ops={os.environ.get('MENDELIHT_XTV_MAX_OPS', 'default')} lanes={os.environ.get('MENDELIHT_CV_LANES', 'default')} world={world}: "
              f"{dt:.3f} s, {launches} fused passes, {ms:.0f} ms in X'R kernels, best k {int(np.argmin(mse)) + 1 if world == 1 else '-'}", flush=True)
np.save(os.path.join(ROOT, "gpurun_out", f"cv100_raw_ops{os.environ.get('MENDELIHT_XTV_MAX_OPS', 'd')}.npy"), ref)
