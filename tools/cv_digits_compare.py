"""BASELINE configs[3] (cv_iht Bernoulli/Logit, path = 1:20, 5 folds, n = 500k, p = 1M) in the default 54-bit residual format
(4910: ten base-49 FP6 digits, three residuals per MFMA operand) and in the 43-bit fast format (4908: eight digits, four per
operand): wall time of the 100-fit cross-validation, and whether any loss, the selected model size or any support moves
(VERDICT r3 item 7).  The supports come from the model path on the full data (iht_run_many_models returns no betas through
the mirror: mih_fit_iht_path is called directly)."""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
from mendeliht_amd import api

n, p = int(os.environ.get("MIH_N", 500_000)), int(os.environ.get("MIH_P", 1_000_000))
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = m.hash_folds(n, 5)
path = list(range(1, 21))
out = {"n": n, "p": p, "workload": "cv_iht Bernoulli/Logit path=1:20 q=5 (BASELINE configs[3])"}
res = {}
for fmt in (0, 4908):
    kw = dict(d=m.Bernoulli(), l=m.LogitLink(), path=path, q=5, folds=folds, verbose=False, return_raw=True, xtv_digits=fmt or None)
    m.cv_iht(yb, x, None, **kw)                       # warm-up
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        mse, raw = m.cv_iht(yb, x, None, **kw)
        ts.append(time.perf_counter() - t0)
    # supports and coefficients of the 20 full-data fits of the model path
    keep = []
    prm = api._params(1, 1, m.Bernoulli(), m.LogitLink(), 1e-4, 100, 5, 3, "None", None, None, None, 1, x.p, keep, xtv_digits=fmt or None)
    pa = np.ascontiguousarray(path, dtype=np.int64)
    z = np.ones((n, 1), order="F")
    logl, iters = np.zeros(pa.size), np.zeros(pa.size, dtype=np.int64)
    betas = np.zeros((pa.size, x.p))
    api._check(api.lib().mih_fit_iht_path(x._h, C.byref(prm), api._p(yb), api._p(z), 1, api._p(pa), pa.size, 0, 1,
                                          api._p(logl), api._p(iters), api._p(betas), None))
    res[fmt] = dict(mse=mse, raw=raw, seconds=ts, logl=logl, iters=iters, betas=betas)
a, b = res[0], res[4908]
out["default_4910"] = {"cv_iht_s": a["seconds"], "best_k": int(np.argmin(a["mse"])) + 1}
out["fast_4908"] = {"cv_iht_s": b["seconds"], "best_k": int(np.argmin(b["mse"])) + 1}
out["speedup_min_over_min"] = min(a["seconds"]) / min(b["seconds"])
out["losses_max_rel_diff"] = float(np.max(np.abs(a["raw"] - b["raw"]) / np.abs(a["raw"])))
out["losses_identical"] = int(np.sum(a["raw"] == b["raw"]))
out["path_supports_differ"] = int(sum(not np.array_equal(np.flatnonzero(a["betas"][i]), np.flatnonzero(b["betas"][i])) for i in range(len(path))))
out["path_iterations_differ"] = int(np.sum(a["iters"] != b["iters"]))
nz = a["betas"] != 0
out["path_beta_max_rel_diff"] = float(np.max(np.abs(a["betas"][nz] - b["betas"][nz]) / np.abs(a["betas"][nz])))
out["path_logl_max_rel_diff"] = float(np.max(np.abs(a["logl"] - b["logl"]) / np.abs(a["logl"])))
print(json.dumps(out))
