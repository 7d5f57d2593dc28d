"""Generate tests/golden/oracle_goldens.json: results of the CPU oracle (after it has reproduced the reference's
recorded run, tests/test_oracle_golden.py) on fixed inputs for the families no reference fixture pins.
Inputs are derived deterministically from the committed fixtures (no RNG): see `scenarios()`.

    python tests/golden/make_oracle_goldens.py        # rewrites the JSON (run only when the oracle changes on purpose)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import FIX, hash_folds          # noqa: E402


def scenarios(ox):
    """Deterministic phenotypes: the linear predictor of the shipped true model (tests/fixtures/normal_true_beta.txt)
    pushed through each family's inverse CDF at u_i = frac(i * 2654435761 / 2^32) -- no RNG state involved.
    name -> (y, oracle fit kwargs, covariates)."""
    from scipy import stats
    n = 1000
    y = np.loadtxt(os.path.join(FIX, "normal_y_fam6.txt"))
    z = np.loadtxt(os.path.join(FIX, "covariates.txt"), delimiter=",")
    z[:, 1:] = (z[:, 1:] - z[:, 1:].mean(axis=0)) / z[:, 1:].std(axis=0, ddof=1)
    tb = [ln.strip().split(",") for ln in open(os.path.join(FIX, "normal_true_beta.txt"))][1:]
    idx = np.array([int(a[3:]) - 1 for a, _ in tb]); val = np.array([float(b) for _, b in tb])
    mask = np.zeros(ox.p, np.uint8); mask[idx] = 1
    coef = np.zeros(ox.p); coef[idx] = val
    eta = ox.xv_masked(mask, coef)
    u = ((np.arange(1, n + 1, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(2 ** 32)).astype(np.float64) / 2 ** 32
    yb = (u < 1 / (1 + np.exp(-0.4 * eta))).astype(float)
    yp = stats.poisson.ppf(u, np.exp(0.25 * eta + 0.3))
    yg = stats.gamma.ppf(u, 4.0, scale=np.exp(0.25 * eta) / 4.0)
    out = {
        "bernoulli_logit_k6": (yb, dict(k=6, dist="bernoulli", link="logit"), None),
        "bernoulli_probit_k6": (yb, dict(k=6, dist="bernoulli", link="probit"), None),
        "poisson_log_k5": (yp, dict(k=5, dist="poisson", link="log"), None),
        "negbin_log_k5_newton": (yp, dict(k=5, dist="negbin", link="log", nb_r=3.0, est_r="newton"), None),
        "negbin_log_k5_mm": (yp, dict(k=5, dist="negbin", link="log", nb_r=3.0, est_r="mm"), None),
        "gamma_log_k4": (yg, dict(k=4, dist="gamma", link="log"), None),
        "normal_cov_zkeep_k8": (y, dict(k=8, zkeep=[1, 0]), z),
        "normal_initbeta_debias_k7": (y, dict(k=7, init_beta=True, debias=True), z),
    }
    return n, out, y, z


def main():
    from oracle import oracle as O
    cols = np.fromfile(os.path.join(FIX, "normal.bed"), dtype=np.uint8)[3:].reshape(-1, 250)
    ox = O.Mat.from_bed_columns(cols, 1000)
    n, sc, y, z = scenarios(ox)
    gold = {}
    for name, (yy, kw, zz) in sc.items():
        o = O.fit_iht(ox, yy, zz, **kw)
        nz = np.flatnonzero(o["beta"])
        # only converged trajectories make goldens; without debias (whose stale gradient makes the next step back off
        # to the limit, as in the reference) also no step that used up its backtracks
        assert o["iter"] < 100 and (kw.get("debias") or o["bt_trace"].max(initial=0) < 3), name
        gold[name] = dict(iter=int(o["iter"]), logl=o["logl"], support=nz.tolist(), beta=o["beta"][nz].tolist(),
                          c=np.asarray(o["c"]).tolist(), backtracks=o["bt_trace"].tolist(), nb_r=o.get("nb_r", 1.0))
    folds = hash_folds(n, 3)
    mse, raw = O.cv_iht(ox, y, z, path=range(1, 9), q=3, folds=folds)
    gold["cv_normal_path1_8_q3"] = dict(mse=mse.tolist(), raw=np.asarray(raw).tolist())
    mcols = np.fromfile(os.path.join(FIX, "multivariate.bed"), dtype=np.uint8)[3:].reshape(-1, 250)
    mx = O.Mat.from_bed_columns(mcols, n)
    Y = np.loadtxt(os.path.join(FIX, "multivariate.phen"), delimiter=",").T.copy()
    om = O.fit_mv(mx, Y, None, k=10)
    nzm = np.argwhere(om["B"] != 0)
    gold["multivariate_k10"] = dict(iter=int(om["iter"]), logl=om["logl"], support=nzm.tolist(),
                                    B=[om["B"][i, j] for i, j in nzm], C=om["C"].tolist(), Sigma=om["Sigma"].tolist())
    with open(os.path.join(HERE, "oracle_goldens.json"), "w") as f:
        json.dump(dict(_source="tests/golden/make_oracle_goldens.py: CPU oracle on deterministic transforms of the committed fixtures",
                       goldens=gold), f, indent=1)
    print("wrote", len(gold), "goldens")


if __name__ == "__main__":
    main()
