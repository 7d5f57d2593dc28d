"""cv_iht wall time on the BASELINE configs[3] geometry: path=1:20, 5 folds, Bernoulli/Logit.
--world W --rank R runs the share of (fold,k) combinations GPU R of W would own."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
hash_folds = m.hash_folds
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=500_000); ap.add_argument("--p", type=int, default=1_000_000)
ap.add_argument("--k", type=int, default=10); ap.add_argument("--world", type=int, default=8); ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--family", default="bernoulli"); ap.add_argument("--kmax", type=int, default=20)
ap.add_argument("--est-r", default="None"); ap.add_argument("--cv-threads", type=int, default=0)
ap.add_argument("--json", default=None, help="append a JSON record of the run to this file")
a = ap.parse_args()
x = m.SnpLinAlg.synthetic(a.n, a.p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(a.p, a.k, replace=False)); beta = rng.standard_normal(a.k) * 0.5
eta = x.xv_sparse(supp, beta)
if a.family == "bernoulli":
    y = (rng.random(a.n) < 1 / (1 + np.exp(-eta))).astype(float); kw = dict(d=m.Bernoulli(), l=m.LogitLink())
elif a.family == "negbin":          # NegBin / Log with the nuisance parameter estimated: chains in lock-step
    mu = np.exp(0.5 + 0.3 * eta)
    y = rng.negative_binomial(5, 5 / (mu + 5)).astype(float)
    kw = dict(d=m.NegativeBinomial(1.0), l=m.LogLink(), est_r=a.est_r, cv_threads=a.cv_threads)
else:
    y = eta + 1 + rng.standard_normal(a.n); kw = {}
folds = hash_folds(a.n, 5)
t0 = time.perf_counter()
mse, raw = m.cv_iht(y, x, None, path=range(1, a.kmax + 1), q=5, folds=folds, verbose=False, return_raw=True,
                    rank=a.rank, world=a.world, **kw)
dt = time.perf_counter() - t0
print(f"cv_iht {a.family} n={a.n} p={a.p} path=1:{a.kmax} q=5 rank {a.rank}/{a.world}: {dt:.2f} s, {np.count_nonzero(raw)} fits")
print("raw losses (nonzero):", np.round(raw[raw != 0][:8], 1))
if a.json:
    import json
    with open(a.json, "a") as f:
        f.write(json.dumps(dict(family=a.family, est_r=a.est_r, cv_threads=a.cv_threads, n=a.n, p=a.p, kmax=a.kmax, rank=a.rank,
                                world=a.world, seconds=dt, fits=int(np.count_nonzero(raw)), best_k=int(np.argmin(mse)) + 1)) + "\n")
