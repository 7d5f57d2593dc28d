"""The allocation path a caller with a small matrix gets (VERDICT r3 "weak" 3).  The rest of the GPU suite asks for the device-memory
reserve of a 125 GB matrix on every test matrix (conftest.py: RESERVE_BY_DEFAULT, mih_mat_reserve) so that the pool / arena /
hand-over code runs in CI; this file runs a representative subset WITHOUT it -- every IHTVariable buffer and lock-step workspace
from hipMalloc, as for any 2-bit matrix under 4 GiB: the reference's recorded run (G1), a GLM-link fit, a cross-validation grid
on the lock-step driver, a multivariate fit, a model path -- all against the oracle -- and the plain-C harness (which never had a
reserve: it calls mih_snp_create directly)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import FIX, GOLD, hash_folds, make_bed

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def library_default_policy():
    import mendeliht_amd.api as api
    old = api.RESERVE_BY_DEFAULT
    api.RESERVE_BY_DEFAULT = False
    yield
    api.RESERVE_BY_DEFAULT = old


def test_recorded_run_g1_without_a_reserve(mih, normal_data):
    """docs/src/man/examples.md:230-267 on the default allocation path."""
    n = normal_data["n"]
    x = mih.SnpLinAlg(mih.read_bed(normal_data["bed"], n), n, center=True, scale=True, impute=True)
    res = mih.fit_iht(normal_data["y"], x, normal_data["z"], k=7, verbose=False)
    g = json.load(open(os.path.join(GOLD, "golden_normal_k7.json")))
    assert res.iter == g["iterations"] and list(res.trace["backtracks"]) == g["backtracks"]
    assert [j + 1 for j in np.flatnonzero(res.beta)] == g["positions_1based"]
    np.testing.assert_allclose(res.trace["logl"], g["logl"], rtol=1e-11)
    np.testing.assert_allclose(res.trace["tol"], g["tol"], rtol=1e-7)
    np.testing.assert_allclose(res.beta[np.flatnonzero(res.beta)], g["beta_printed"], rtol=5e-6)
    np.testing.assert_allclose(res.c, g["c_printed"], rtol=5e-6)


def _problem(mih, oracle, seed, n=1300, p=420, miss=0.02):
    rng = np.random.default_rng(seed)
    cols = make_bed(rng, n, p, missing_rate=miss)
    x = mih.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)
    ox = oracle.Mat.from_bed_columns(cols, n)
    supp = np.sort(rng.choice(p, 6, replace=False))
    mask = np.zeros(p, np.uint8); mask[supp] = 1
    b = np.zeros(p); b[supp] = rng.standard_normal(6) * 0.6
    eta = ox.xv_masked(mask, b)
    return rng, x, ox, eta, n, p


def test_family_fit_and_cv_grid_without_a_reserve(mih, oracle):
    """Bernoulli/Logit fit, then a 4 x 9 cross-validation (36 fits: two lock-step lanes, the tail hand-over) and a model path,
    every IHTVariable and fused-pass workspace allocated with hipMalloc -- against the oracle."""
    rng, x, ox, eta, n, p = _problem(mih, oracle, 404)
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    res = mih.fit_iht(yb, x, None, k=6, d=mih.Bernoulli(), l=mih.LogitLink(), verbose=False)
    o = oracle.fit_iht(ox, yb, None, k=6, dist="bernoulli", link="logit")
    assert res.iter == o["iter"] and np.array_equal(np.flatnonzero(res.beta), np.flatnonzero(o["beta"]))
    np.testing.assert_allclose(res.beta, o["beta"], rtol=1e-4, atol=1e-12)
    assert res.logl == pytest.approx(o["logl"], rel=1e-9)
    folds = hash_folds(n, 4)
    path = list(range(1, 10))
    mih.profile_counters(x, reset=True)
    mih.profile_enable(x, True)
    mse, raw = mih.cv_iht(yb, x, None, d=mih.Bernoulli(), l=mih.LogitLink(), path=path, q=4, folds=folds, verbose=False, return_raw=True)
    mih.profile_enable(x, False)
    cnt = mih.profile_counters(x, reset=True)
    assert cnt["fits"] == 36 and cnt["lanes"] == 2
    omse, oraw = oracle.cv_iht(ox, yb, None, path=path, q=4, folds=folds, dist="bernoulli", link="logit")
    np.testing.assert_allclose(raw, oraw, rtol=1e-7)
    np.testing.assert_allclose(mse, omse, rtol=1e-7)
    yn = eta + 1 + rng.standard_normal(n)
    ll = mih.iht_run_many_models(yn, x, None, path=[2, 4, 6, 8], verbose=False)
    for kk, got in zip([2, 4, 6, 8], ll):
        assert got == pytest.approx(oracle.fit_iht(ox, yn, None, k=kk, max_iter=100)["logl"], rel=1e-9)


def test_multivariate_fit_without_a_reserve(mih, oracle):
    rng, x, ox, eta, n, p = _problem(mih, oracle, 405)
    r = 3
    B = np.zeros((r, p))
    for t in range(r):
        B[t, rng.choice(p, 3, replace=False)] = rng.standard_normal(3) * 0.5
    XB = np.stack([ox.xv_masked((B[t] != 0).astype(np.uint8), B[t]) for t in range(r)])
    Z = np.vstack([np.ones(n), rng.standard_normal(n)])
    Y = XB + np.array([[0.5], [-0.3], [0.1]]) + 0.4 * rng.standard_normal((r, n))
    res = mih.fit_iht(Y, x, Z, k=9, verbose=False)
    o = oracle.fit_mv(ox, Y, Z, k=9)
    assert res.iter == o["iter"] and np.array_equal(res.beta != 0, o["B"] != 0)
    np.testing.assert_allclose(res.beta, o["B"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.Σ, o["Sigma"], rtol=1e-8)
    folds = hash_folds(n, 3)
    mse = mih.cv_iht(Y, x, Z, path=[3, 6, 9], q=3, folds=folds, verbose=False)
    omse, _ = oracle.cv_mv(ox, Y, Z, path=[3, 6, 9], q=3, folds=folds)
    np.testing.assert_allclose(mse, omse, rtol=1e-6)


def test_c_harness_runs_on_the_default_path(mih, tmp_path):
    """tests/abi_harness.c calls mih_snp_create itself: no reserve is ever asked for (the library reads no environment switch
    that could add one -- tests/test_abi_cpu.py)."""
    from test_abi_cpu import _build_harness
    exe = _build_harness(tmp_path)
    env = {k: v for k, v in os.environ.items() if not k.startswith("MENDELIHT_")}
    r = subprocess.run([str(exe), mih.library_path(), FIX], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS" in r.stdout and "5 iterations" in r.stdout
