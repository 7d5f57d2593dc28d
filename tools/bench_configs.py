"""All BASELINE.json configs beside the bench.py headline (configs[2]) in one run; writes one JSON object.

  configs[1]  dense Float64 50 000 x 100 000, k=100, Normal          (X'r GB/s, ms/iteration)
  configs[3]  SnpArray 500k x 1M, Bernoulli/Logit, cv_iht path=1:20, 5 folds: the 13-fit share of one of
              8 GPUs, and all 100 fits on this single GPU
  configs[4]  MvNormal r=10 traits, k=500 on the same matrix          (X'R pass, ms/iteration)
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
hash_folds = m.hash_folds

digits = int(os.environ.get("MIH_DIGITS", 0))      # xtv_digits: 0 default, 4908 = the opt-in fast mode for fused multi-RHS passes
m.set_xtv_digits(digits)
out = {"xtv_digits": digits}
# ---- configs[1]
n, p, k = 50_000, 100_000, 100
x = m.DenseMatrix.synthetic(n, p, seed=7)
ms, _ = x.bench_xtv(-1, iters=10, warmup=2)
rng = np.random.default_rng(1)
supp = np.sort(rng.choice(p, k, replace=False))
y = x.xv_sparse(supp, rng.standard_normal(k)) + 1 + rng.standard_normal(n)
res = m.fit_iht(y, x, None, k=k, verbose=False)
out["configs[1] dense f64 50000x100000 k=100"] = dict(
    xtv_ms=ms, xtv_GBps=x.algorithmic_bytes() / ms / 1e6, iterations=int(res.iter), ms_per_iteration=1e3 * res.time / res.iter,
    recovered=f"{np.intersect1d(np.flatnonzero(res.beta), supp).size}/{k}")
print(json.dumps(out), flush=True)
del x

# ---- configs[3] and [4] share the 125 GB matrix
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = hash_folds(n, 5)
cv = {}
for world, label in ((8, "share of GPU 0 of 8 (13 fits)"), (1, "all 100 fits on one GPU")):
    t0 = time.perf_counter()
    mse, raw = m.cv_iht(yb, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True,
                        rank=0, world=world, d=m.Bernoulli(), l=m.LogitLink())
    cv[label] = dict(seconds=time.perf_counter() - t0, fits=int(np.count_nonzero(raw)))
    if world == 1:
        cv["best_k"] = int(np.argmin(mse)) + 1
out["configs[3] cv_iht Bernoulli/Logit path=1:20 q=5 n=500k p=1M"] = cv
print(json.dumps(out), flush=True)

r, k = 10, 500
ms10, _ = x.bench_xtv_batched(r, max_fused=4, iters=3, warmup=1)
rng = np.random.default_rng(3)
lin = rng.choice(r * p, k, replace=False)
B = {}
Y = rng.standard_normal((r, n))
for t in range(r):
    cols = np.sort(lin[lin % r == t] // r)
    cols = np.unique(cols)
    Y[t] += x.xv_sparse(cols, rng.standard_normal(cols.size) * 0.3) + 1.0
t0 = time.perf_counter()
m.profile_read(x, reset=True); m.profile_enable(x, True)
res = m.fit_iht(Y, x, None, k=k, verbose=False, max_iter=8)
wall = time.perf_counter() - t0
m.profile_enable(x, False)
ps = m.profile_passes(x, reset=True)                 # [0] = the initial score; HIP events around every pass kernel
gaps = [ps[i + 1]["start_ms"] - ps[i]["start_ms"] - ps[i]["ms"] for i in range(1, len(ps) - 1)]
out["configs[4] MvNormal r=10 k=500 n=500k p=1M"] = dict(
    xtR_ms=ms10, flop_equiv_TFLOPs=2.0 * n * p * r / (ms10 * 1e-3) / 1e12, iterations=int(res.iter),
    ms_per_iteration=1e3 * res.time / res.iter, wall_s=wall, nonzero=int(np.count_nonzero(res.beta)),
    pass_ms_in_the_fit=sum(q["ms"] for q in ps[1:]) / max(len(ps) - 1, 1),
    outside_the_pass_ms=sum(gaps) / max(len(gaps), 1),      # end of a step's pass -> start of the next step's pass, steady state
    note="outside_the_pass_ms is measured between the HIP events of consecutive pass kernels; ms_per_iteration = fit time / iterations also carries the fit's end (the last score, which nothing waits for until the fit returns)")
print(json.dumps(out), flush=True)
with open(os.path.join(ROOT, "gpurun_out", f"configs_digits{digits}.json"), "w") as f:
    json.dump(out, f, indent=1)
