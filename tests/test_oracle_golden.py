"""The CPU oracle against the reference's own recorded outputs and invariants (no GPU)."""
import json
import os

import numpy as np
import pytest
from scipy import stats

from conftest import FIX, GOLD, check_recorded_cv_curve, hash_folds, make_bed, perm_folds


def test_g1_recorded_log_normal_k7(oracle, normal_data):
    """docs/src/man/examples.md:230-267: iht("normal", 7, Normal, covariates=..., phenotypes=6)."""
    g = json.load(open(os.path.join(GOLD, "golden_normal_k7.json")))
    x = oracle.Mat.from_bed_file(normal_data["bed"], normal_data["n"])
    r = oracle.fit_iht(x, normal_data["y"], normal_data["z"], k=7)
    assert r["iter"] == g["iterations"]
    np.testing.assert_allclose(r["logl_trace"], g["logl"], rtol=1e-12)
    np.testing.assert_allclose(r["tol_trace"], g["tol"], rtol=1e-9)
    assert list(r["bt_trace"]) == g["backtracks"]
    assert r["logl"] == pytest.approx(g["final_logl"], rel=1e-13)
    nz = np.flatnonzero(r["beta"])
    assert list(nz + 1) == g["positions_1based"]          # bit-exact support
    np.testing.assert_allclose(r["beta"][nz], g["beta_printed"], rtol=5e-6)   # printed to 6 digits
    np.testing.assert_allclose(r["c"], g["c_printed"], rtol=5e-6)
    assert r["pve"] == pytest.approx(g["pve"], rel=1e-10)
    assert not r["choose_fired"]


def test_g2_shipped_summary_k8(oracle, normal_data):
    """data/iht.summary.txt (older release): converged optimum only."""
    g = json.load(open(os.path.join(GOLD, "golden_iht_summary_k8.json")))
    x = oracle.Mat.from_bed_file(normal_data["bed"], normal_data["n"])
    r = oracle.fit_iht(x, normal_data["y2"], normal_data["z"], k=8)
    nz = np.flatnonzero(r["beta"])
    assert list(nz + 1) == g["positions_1based"]
    np.testing.assert_allclose(r["beta"][nz], g["beta_printed"], rtol=2e-4)
    np.testing.assert_allclose(r["c"], g["c_printed"], rtol=1e-4)
    assert r["logl"] == pytest.approx(g["final_logl"], rel=1e-6)
    assert r["pve"] == pytest.approx(g["pve"], rel=1e-4)


def test_g1b_k9_intercept_only(oracle, normal_data):
    """BASELINE config 1 (README.md:104 gives the command, no output): SURVEY's independent probe."""
    x = oracle.Mat.from_bed_file(normal_data["bed"], normal_data["n"])
    r = oracle.fit_iht(x, normal_data["y"], None, k=9)
    assert r["iter"] == 10
    assert r["logl"] == pytest.approx(-1612.734968, abs=1e-5)
    assert list(np.flatnonzero(r["beta"]) + 1) == [1266, 3137, 4246, 4717, 6290, 7629, 7755, 8375, 9415]
    assert r["c"][0] == pytest.approx(1.65222721, abs=1e-7)


@pytest.mark.parametrize("curve", ["docs_curve", "shipped_summary_curve"])
def test_cv_curves_the_reference_recorded(oracle, normal_data, curve):
    """cross_validate on the shipped data against the two curves the reference itself holds (docs/src/man/examples.md:169-192,
    data/cviht.summary.txt): the only reference-held evidence for predict!'s deviance sum and meanloss's fold weights
    (cross_validation.jl:279-286, 304-320).  The reference's folds are random and unrecorded: three explicit fold seeds each."""
    gold = json.load(open(os.path.join(GOLD, "golden_cv_normal.json")))
    g = gold[curve]
    x = oracle.Mat.from_bed_file(normal_data["bed"], normal_data["n"])
    y = normal_data["y"] if g["y"] == "normal_y_fam6.txt" else normal_data["y2"]
    for seed in gold["fold_seeds"]:
        folds = perm_folds(normal_data["n"], g["q"], seed)
        mse, raw = oracle.cv_iht(x, y, normal_data["z"], path=g["path"], q=g["q"], folds=folds, zkeep=g["zkeep"])
        check_recorded_cv_curve(mse, g)
        # meanloss (cross_validation.jl:312-317): fold j's deviance sum weighted by its share of the samples
        wts = np.bincount(folds, minlength=g["q"] + 1)[1:] / normal_data["n"]
        np.testing.assert_allclose(mse, wts @ raw, rtol=1e-13)


def test_snplinalg_semantics(oracle):
    """mu = mean of non-missing, sinv = 1/sqrt(mu(1-mu/2)) else 1; X'r against a dense numpy matrix."""
    rng = np.random.default_rng(0)
    n, p = 203, 57
    cols = make_bed(rng, n, p, missing_rate=0.05)
    cols[3, :] = 0                      # monomorphic SNP -> sinv = 1
    x = oracle.Mat.from_bed_columns(cols, n)
    code = np.stack([(cols[:, i // 4] >> (2 * (i % 4))) & 3 for i in range(n)], axis=1)   # p x n
    g = np.where(code == 2, 1.0, np.where(code == 3, 2.0, 0.0))
    miss = code == 1
    mu = g.sum(1) / (~miss).sum(1)
    sd = np.sqrt(mu * (1 - mu / 2))
    sinv = np.where(sd > 0, 1 / np.where(sd > 0, sd, 1), 1.0)
    omu, osinv = x.mu_sinv()
    np.testing.assert_allclose(omu, mu, rtol=1e-14)
    np.testing.assert_allclose(osinv, sinv, rtol=1e-14)
    assert osinv[3] == 1.0
    X = ((np.where(miss, mu[:, None], g) - mu[:, None]) * sinv[:, None]).T     # n x p standardized, imputed
    r = rng.standard_normal(n)
    np.testing.assert_allclose(x.xtv(r), X.T @ r, rtol=1e-11, atol=1e-11)
    idx = np.zeros(p, np.uint8)
    idx[[1, 5, 40]] = 1
    coef = rng.standard_normal(p)
    sel = idx.astype(bool)
    np.testing.assert_allclose(x.xv_masked(idx, coef), X[:, sel] @ coef[sel], rtol=1e-11, atol=1e-12)
    assert x.getindex(7, 5) == pytest.approx(X[7, 5], rel=1e-14, abs=1e-15)


def test_blocked_xtv_equals_column_loop(oracle):
    """orc_xtv walks column blocks over row tiles for cache reuse; every column still adds its rows in the same
    order, so it must agree bit for bit with the plain column-at-a-time loop (ragged n, missing data)."""
    rng = np.random.default_rng(12)
    for n, p in ((1003, 77), (9001, 130), (20000, 33)):
        ox = oracle.Mat.from_bed_columns(make_bed(rng, n, p, missing_rate=0.03), n)
        r = rng.standard_normal(n)
        assert np.array_equal(ox.xtv(r), ox.xtv_colwise(r))


def test_loglikelihood_vs_logpdf(oracle):
    """test/utilities_test.jl:20-51: loglikelihood equals the sum of logpdfs; :53-61 deviance."""
    rng = np.random.default_rng(1)
    n = 500
    w = np.ones(n)
    L, p_ = oracle.lib(), oracle._p
    mu = rng.uniform(0.05, 0.95, n)
    y = (rng.random(n) < mu).astype(float)
    ll = L.orc_loglikelihood(oracle.BERNOULLI, 1.0, p_(y), p_(mu), p_(w), n)
    assert ll == pytest.approx(stats.bernoulli.logpmf(y, mu).sum(), rel=1e-12)
    mu = rng.uniform(0.5, 5, n)
    y = rng.poisson(mu).astype(float)
    ll = L.orc_loglikelihood(oracle.POISSON, 1.0, p_(y), p_(mu), p_(w), n)
    assert ll == pytest.approx(stats.poisson.logpmf(y, mu).sum(), rel=1e-12)
    rr = 3.5
    y = rng.negative_binomial(rr, rr / (mu + rr)).astype(float)
    ll = L.orc_loglikelihood(oracle.NEGBIN, rr, p_(y), p_(mu), p_(w), n)
    assert ll == pytest.approx(stats.nbinom.logpmf(y, rr, rr / (mu + rr)).sum(), rel=1e-10)
    mu = rng.standard_normal(n)
    y = mu + rng.standard_normal(n)
    ll = L.orc_loglikelihood(oracle.NORMAL, 1.0, p_(y), p_(mu), p_(w), n)
    sd = np.sqrt(((y - mu) ** 2).sum() / n)
    assert ll == pytest.approx(stats.norm.logpdf(y, mu, sd).sum(), rel=1e-12)
    dev = L.orc_deviance(oracle.NORMAL, 1.0, p_(y), p_(mu), p_(w), n)
    assert dev == pytest.approx(((y - mu) ** 2).sum(), rel=1e-14)
    # Gamma(1/phi, mu*phi) and InverseGaussian(mu, 1/phi) with phi = deviance / n (utilities.jl:15, 34-35)
    mu = rng.uniform(0.5, 3.0, n)
    y = rng.gamma(4.0, mu / 4.0)
    dev = L.orc_deviance(oracle.GAMMA, 1.0, p_(y), p_(mu), p_(w), n)
    assert dev == pytest.approx((-2 * (np.log(y / mu) - (y - mu) / mu)).sum(), rel=1e-13)
    phi = dev / n
    ll = L.orc_loglikelihood(oracle.GAMMA, 1.0, p_(y), p_(mu), p_(w), n)
    assert ll == pytest.approx(stats.gamma.logpdf(y, 1 / phi, scale=mu * phi).sum(), rel=1e-11)
    y = rng.wald(mu, 5.0)
    dev = L.orc_deviance(oracle.INVGAUSS, 1.0, p_(y), p_(mu), p_(w), n)
    assert dev == pytest.approx(((y - mu) ** 2 / (y * mu ** 2)).sum(), rel=1e-13)
    lam = n / dev
    ll = L.orc_loglikelihood(oracle.INVGAUSS, 1.0, p_(y), p_(mu), p_(w), n)
    assert ll == pytest.approx(stats.invgauss.logpdf(y, mu / lam, scale=lam).sum(), rel=1e-11)


def test_links(oracle):
    """test/utilities_test.jl:63-92."""
    L = oracle.lib()
    for eta in (-3.0, -0.2, 0.0, 1.7):
        assert L.orc_linkinv(oracle.IDENTITY, eta) == eta
        assert L.orc_linkinv(oracle.LOGIT, eta) == pytest.approx(1 / (1 + np.exp(-eta)), rel=1e-15)
        assert L.orc_linkinv(oracle.LOG, eta) == pytest.approx(np.exp(eta), rel=1e-15)
        mu = 1 / (1 + np.exp(-eta))
        assert L.orc_mueta(oracle.LOGIT, eta) == pytest.approx(mu * (1 - mu), rel=1e-13)
        assert L.orc_linkinv(oracle.PROBIT, eta) == pytest.approx(stats.norm.cdf(eta), rel=1e-13)
        assert L.orc_linkinv(oracle.CLOGLOG, eta) == pytest.approx(1 - np.exp(-np.exp(eta)), rel=1e-13)
        assert L.orc_linkinv(oracle.CAUCHIT, eta) == pytest.approx(stats.cauchy.cdf(eta), rel=1e-13)
        assert L.orc_linkinv(oracle.SQRT, eta) == eta * eta
    for eta in (0.3, 1.0, 2.5):                              # links with a restricted domain
        assert L.orc_linkinv(oracle.INVERSE, eta) == 1 / eta
        assert L.orc_linkinv(oracle.INVSQUARE, eta) == pytest.approx(eta ** -0.5, rel=1e-15)
    # mueta is the derivative of linkinv for every link (GLM.jl glmtools.jl)
    for link in range(9):
        for eta in (0.4, 1.3):
            h = 1e-6
            fd = (L.orc_linkinv(link, eta + h) - L.orc_linkinv(link, eta - h)) / (2 * h)
            assert L.orc_mueta(link, eta) == pytest.approx(fd, rel=1e-7), link


def test_project_k_property(oracle):
    """test/utilities_test.jl:166-176."""
    rng = np.random.default_rng(2)
    x = rng.random(100000)
    k = 100
    out = oracle.project_k(x, k)
    keep = np.argsort(-x)[:k]
    assert np.count_nonzero(out) == k
    assert np.array_equal(np.sort(np.flatnonzero(out)), np.sort(keep))
    assert np.array_equal(out[keep], x[keep])
    t = oracle.project_k(np.array([1.0, -2.0, 2.0, 0.5]), 2)      # ties at the threshold are kept
    assert list(t) == [0.0, -2.0, 2.0, 0.0]
    t = oracle.project_k(np.array([1.0, -2.0, 2.0, 0.5]), 1)
    assert list(t) == [0.0, -2.0, 2.0, 0.0]
    with pytest.raises(ValueError):
        oracle.project_k(x, -1)


def test_project_group_sparse_properties(oracle):
    """test/utilities_test.jl:180-213."""
    rng = np.random.default_rng(3)
    m, n, k, J = 5, 50, 3, 2
    y = rng.standard_normal(n)
    group = np.repeat(np.arange(1, m + 1), n // m)
    out = oracle.project_group_sparse(y, group, J, k)
    nzg = [np.count_nonzero(out[group == g]) for g in range(1, m + 1)]
    assert sum(c > 0 for c in nzg) == J
    assert all(c in (0, k) for c in nzg)
    assert np.count_nonzero(out) == J * k
    one = np.ones(n, dtype=np.int64)                      # one group == project_k
    assert np.array_equal(oracle.project_group_sparse(y, one, 1, 7), oracle.project_k(y, 7))
    ks = np.array([1, 2, 3, 4, 5])                        # per-group k vector
    out = oracle.project_group_sparse(y, group, 5, ks)
    assert [np.count_nonzero(out[group == g]) for g in range(1, m + 1)] == list(ks)


def test_fit_invariants_per_family(oracle):
    """test/L0_reg_test.jl:21-24,49-52,74-77,99-102: count(!iszero, beta) == k, intercept estimated."""
    rng = np.random.default_rng(4)
    n, p, k = 600, 800, 6
    cols = make_bed(rng, n, p)
    x = oracle.Mat.from_bed_columns(cols, n)
    b = np.zeros(p)
    supp = rng.choice(p, k, replace=False)
    b[supp] = rng.choice([-1, 1], k) * rng.uniform(0.3, 0.8, k)
    m = np.zeros(p, np.uint8)
    m[supp] = 1
    eta = x.xv_masked(m, b)
    cases = [("normal", "identity", eta + 1 + rng.standard_normal(n)),
             ("bernoulli", "logit", (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)),
             ("poisson", "log", rng.poisson(np.exp(0.4 * eta)).astype(float)),
             ("negbin", "log", rng.negative_binomial(10, 10 / (np.exp(0.4 * eta) + 10)).astype(float))]
    for dist, link, y in cases:
        r = oracle.fit_iht(x, y, None, k=k, dist=dist, link=link, nb_r=10.0)
        assert np.count_nonzero(r["beta"]) == k, dist
        assert r["c"][0] != 0
        assert np.isfinite(r["logl"])


def test_zkeep_and_cv(oracle):
    """test/L0_reg_test.jl:169-173 (zkeep) and test/cv_iht_test.jl:29-38 (mses > 0; k > p throws)."""
    rng = np.random.default_rng(5)
    n, p = 400, 300
    cols = make_bed(rng, n, p)
    x = oracle.Mat.from_bed_columns(cols, n)
    z = np.column_stack([np.ones(n), rng.standard_normal(n), rng.standard_normal(n)])
    b = np.zeros(p)
    b[[5, 50, 200]] = [0.7, -0.6, 0.5]
    m = (b != 0).astype(np.uint8)
    y = x.xv_masked(m, b) + z @ np.array([1.0, 1.5, 0.0]) + rng.standard_normal(n)
    r = oracle.fit_iht(x, y, z, k=4, zkeep=[1, 0, 0])
    assert r["c"][0] != 0
    assert np.count_nonzero(r["beta"]) + np.count_nonzero(r["c"][1:]) == 4
    folds = hash_folds(n, 3)
    mse, raw = oracle.cv_iht(x, y, z, path=range(0, 8), q=3, folds=folds, max_iter=10)
    assert mse.shape == (8,) and np.all(mse > 0) and np.all(raw > 0)
    with pytest.raises(RuntimeError):
        oracle.cv_iht(x, y, z, path=[p + 1], q=3, folds=folds)


def test_multivariate_shipped_data(oracle):
    """SURVEY 8c G3 plausibility target: data/multivariate.* with k=10 (true Sigma is shipped)."""
    n = 1000
    x = oracle.Mat.from_bed_file(os.path.join(FIX, "multivariate.bed"), n)
    Y = np.loadtxt(os.path.join(FIX, "multivariate.phen"), delimiter=",").T
    S = np.loadtxt(os.path.join(FIX, "multivariate.trait.cov"), delimiter=",")
    r = oracle.fit_mv(x, Y, None, k=10)
    assert r["iter"] >= 5 and np.isfinite(r["logl"])
    assert np.count_nonzero(r["B"]) <= 10
    np.testing.assert_allclose(r["Sigma"], S, atol=0.12)
    assert np.all(r["pve"] > 0)
    sel0 = set(np.flatnonzero(r["B"][0]) + 1)
    sel1 = set(np.flatnonzero(r["B"][1]) + 1)
    assert {134, 442, 450, 1891, 2557, 3243} <= sel0 and {1014, 5214} <= sel1
    # independent restatement of what the fit returns: Sigma = R R' / n at the returned model (solve_Sigma!, multivariate.jl:276-282)
    # and logl = the multivariate-normal loglikelihood of the residuals without its 2 pi constant (multivariate.jl:9-13), by scipy
    B, Cm = r["B"], r["C"]
    XB = np.vstack([x.xv_masked((B[i] != 0).astype(np.uint8), B[i]) for i in range(2)])
    R = Y - XB - Cm @ np.ones((1, n))
    np.testing.assert_allclose(R @ R.T / n, r["Sigma"], rtol=0, atol=1e-13)
    want = stats.multivariate_normal(mean=np.zeros(2), cov=r["Sigma"]).logpdf(R.T).sum() + n * 2 / 2 * np.log(2 * np.pi)
    assert r["logl"] == pytest.approx(want, rel=1e-12)


def _oracle_goldens():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_oracle_goldens", os.path.join(GOLD, "make_oracle_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, json.load(open(os.path.join(GOLD, "oracle_goldens.json")))["goldens"]


def test_oracle_reproduces_its_committed_goldens(oracle):
    """tests/golden/oracle_goldens.json was generated by this oracle (make_oracle_goldens.py) for the families no
    reference fixture pins; regenerating it must give the same numbers (guards the oracle against silent drift)."""
    mod, gold = _oracle_goldens()
    cols = np.fromfile(os.path.join(FIX, "normal.bed"), dtype=np.uint8)[3:].reshape(-1, 250)
    ox = oracle.Mat.from_bed_columns(cols, 1000)
    n, sc, y, z = mod.scenarios(ox)
    assert set(sc) <= set(gold)
    for name, (yy, kw, zz) in sc.items():
        o = oracle.fit_iht(ox, yy, zz, **kw)
        g = gold[name]
        assert o["iter"] == g["iter"] and list(o["bt_trace"]) == g["backtracks"], name
        assert list(np.flatnonzero(o["beta"])) == g["support"], name
        np.testing.assert_allclose(o["beta"][g["support"]], g["beta"], rtol=1e-10, err_msg=name)
        assert o["logl"] == pytest.approx(g["logl"], rel=1e-12), name
    mse, _ = oracle.cv_iht(ox, y, z, path=range(1, 9), q=3, folds=hash_folds(n, 3))
    np.testing.assert_allclose(mse, gold["cv_normal_path1_8_q3"]["mse"], rtol=1e-10)


def test_glm_refit_of_debias_against_scikit_learn(oracle):
    """debias! refits the support by GLM (utilities.jl:1014-1020); GLM.jl is not vendored in the reference, so the oracle RESTATES
    its IRLS (mustart, working response, step halving, rtol = atol = 1e-6 on the deviance).  Pin that restatement against an
    independent implementation: scikit-learn's unpenalised GLM solvers converge to the same maximum-likelihood coefficients -- to
    the 1e-6 the IRLS stopping rule allows (Normal/Identity, one exact WLS step: 1e-10 against numpy's least squares)."""
    from sklearn.linear_model import GammaRegressor, LogisticRegression, PoissonRegressor
    rng = np.random.default_rng(2026)
    n, p = 5000, 40
    X = rng.standard_normal((n, p)) * 0.7
    ox = oracle.Mat.from_dense(np.asfortranarray(X))
    mask = np.zeros(p, np.uint8)
    supp = np.sort(rng.choice(p, 7, replace=False))
    mask[supp] = 1
    beta = rng.standard_normal(7) * 0.4
    eta = X[:, supp] @ beta
    Xs = X[:, supp]
    # Normal / Identity: ordinary least squares
    y = eta + rng.standard_normal(n)
    b = oracle.debias_glm(ox, mask, y, "normal", "identity")
    assert np.all(b[mask == 0] == 0)
    np.testing.assert_allclose(b[supp], np.linalg.lstsq(Xs, y, rcond=None)[0], rtol=1e-10, atol=1e-12)
    # Bernoulli / Logit
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    b = oracle.debias_glm(ox, mask, yb, "bernoulli", "logit")
    ref = LogisticRegression(penalty=None, fit_intercept=False, tol=1e-12, max_iter=2000).fit(Xs, yb).coef_.ravel()
    np.testing.assert_allclose(b[supp], ref, rtol=0, atol=2e-6)
    # Poisson / Log
    yp = rng.poisson(np.exp(0.5 * eta)).astype(float)
    b = oracle.debias_glm(ox, mask, yp, "poisson", "log")
    ref = PoissonRegressor(alpha=0, fit_intercept=False, tol=1e-12, max_iter=2000).fit(Xs, yp).coef_
    np.testing.assert_allclose(b[supp], ref, rtol=0, atol=2e-6)
    # Gamma / Log (the coefficients of a Gamma GLM do not depend on the dispersion)
    yg = rng.gamma(2.0, np.exp(0.5 * eta) / 2.0)
    b = oracle.debias_glm(ox, mask, yg, "gamma", "log")
    ref = GammaRegressor(alpha=0, fit_intercept=False, tol=1e-12, max_iter=2000).fit(Xs, yg).coef_
    np.testing.assert_allclose(b[supp], ref, rtol=0, atol=5e-5)        # the deviance of a Gamma fit is flat near the optimum: the 1e-6 stopping rule leaves ~1e-5 in beta


def test_score_is_the_gradient_of_the_loglikelihood(oracle):
    """score! (utilities.jl:126-135) forms X'r with r_i = mu'(eta_i) / Var(mu_i) * (y_i - mu_i); for every family / link pair the
    path supports that must be the gradient of loglikelihood (utilities.jl:9-43) with respect to beta (for the dispersion
    families: of the loglikelihood at fixed phi, times phi).  The reference checks its multivariate gradient against ForwardDiff
    in a notebook (test/multivariate_gradient.ipynb); this pins the restated GLM.jl / Distributions.jl closed forms -- linkinv,
    mueta, glmvar, loglik_obs -- against each other by central differences, independently of any of them being right by itself."""
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(7)
    n, p = 60, 5
    X = rng.standard_normal((n, p)) * 0.5
    beta = rng.standard_normal(p) * 0.3
    DIST = {"normal": 0, "bernoulli": 1, "poisson": 2, "negbin": 3, "gamma": 4, "invgauss": 5}
    LINK = {"identity": 0, "logit": 1, "log": 2, "probit": 3, "cloglog": 4, "cauchit": 5, "inverse": 6, "invsquare": 7, "sqrt": 8}
    cases = [("bernoulli", "logit"), ("bernoulli", "probit"), ("bernoulli", "cloglog"), ("bernoulli", "cauchit"),
             ("poisson", "log"), ("poisson", "sqrt"), ("negbin", "log"), ("normal", "identity"), ("gamma", "log"), ("invgauss", "log")]
    nb_r = 3.5
    for dist, link in cases:
        d, l = DIST[dist], LINK[link]
        off = 2.0 if link in ("sqrt",) else 0.0                        # keep sqrt-link means away from 0
        eta0 = X @ beta + off
        mu0 = np.array([L.orc_linkinv(l, float(e)) for e in eta0])
        if dist == "bernoulli":
            y = (rng.random(n) < mu0).astype(float)
        elif dist in ("poisson", "negbin"):
            y = rng.poisson(mu0).astype(float)
        elif dist == "normal":
            y = mu0 + rng.standard_normal(n)
        else:
            y = rng.gamma(2.0, mu0 / 2.0)
        w = np.ones(n)

        def mu_of(b):
            return np.array([L.orc_linkinv(l, float(e)) for e in X @ b + off])

        phi = L.orc_deviance(d, nb_r, y.ctypes.data_as(C.c_void_p), mu0.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), n) / n

        def loglik(b):                                                   # sum_i loglik_obs(d, y_i, mu_i, 1, phi) at FIXED phi
            m = mu_of(b)
            return sum(L.orc_loglik_obs(d, float(y[i]), float(m[i]), 1.0, float(phi), nb_r) for i in range(n))

        r = np.array([L.orc_mueta(l, float(eta0[i])) / L.orc_glmvar(d, float(mu0[i]), nb_r) * (y[i] - mu0[i]) for i in range(n)])
        score = X.T @ r
        h = 1e-6
        num = np.array([(loglik(beta + h * np.eye(p)[j]) - loglik(beta - h * np.eye(p)[j])) / (2 * h) for j in range(p)])
        # dispersion families: d loglik / d eta = r / phi' with phi' = phi (Normal), phi (Gamma: shape 1/phi), phi (InverseGaussian)
        scale = phi if dist in ("normal", "gamma", "invgauss") else 1.0
        np.testing.assert_allclose(num * scale, score, rtol=2e-5, atol=1e-6, err_msg=f"{dist}/{link}")


def test_negbin_nuisance_parameter_against_scipy(oracle):
    """mle_for_r (utilities.jl:141-247): the Newton update (digamma / trigamma score and curvature, line search on the
    loglikelihood) must end at the maximiser over r of sum_i logpmf(NegativeBinomial(r, r / (mu_i + r)), y_i) -- found here
    independently by scipy's bounded scalar minimiser on scipy.stats.nbinom."""
    from scipy import optimize
    rng = np.random.default_rng(11)
    n = 4000
    mu = np.exp(rng.normal(0.8, 0.5, n))
    for r_true in (1.5, 4.0, 12.0):
        y = rng.negative_binomial(r_true, r_true / (mu + r_true)).astype(float)

        def nll(r):
            return -stats.nbinom.logpmf(y, r, r / (mu + r)).sum()
        best = optimize.minimize_scalar(nll, bounds=(1e-3, 1e3), method="bounded", options={"xatol": 1e-10})
        r_newton = oracle.mle_for_r(y, mu, r0=1.0, method="newton")
        assert r_newton == pytest.approx(best.x, rel=2e-5), r_true            # stops at |dr| <= 1e-6 (utilities.jl:242)
        assert abs(r_newton - r_true) < 0.25 * r_true                           # and it is the right quantity
        # :MM is ONE update r <- -sum_i sum_{j<y_i} r/(r+j) / sum_i log(r/(r+mu_i)) (utilities.jl:158-173).  With mu held fixed its fixed
        # point is not the maximiser (the term sum (mu_i - y_i)/(mu_i + r) of the score is missing: it vanishes only for the
        # intercept-only mean) -- that is the reference's update, so it is pinned as a formula, not as an optimiser
        r0 = 2.0
        num = sum((r0 / (r0 + np.arange(int(v)))).sum() for v in y)
        den = np.log(r0 / (r0 + mu)).sum()
        assert oracle.mle_for_r(y, mu, r0=r0, method="mm") == pytest.approx(-num / den, rel=1e-12)


def test_project_group_sparse_against_a_set_based_statement(oracle):
    """project_group_sparse! (utilities.jl:613-679) is written as two passes over a sort permutation.  The same projection stated
    as sets, in numpy: within each group keep its k_g largest |y| (k_g scalar or per group); rank the groups by the squared norm
    of what they kept; keep the J best groups, zero everything else.  Continuous random data (no ties): the two must agree."""
    rng = np.random.default_rng(613)
    for trial in range(20):
        p = int(rng.integers(30, 400))
        G = int(rng.integers(1, 12))
        group = rng.integers(1, G + 1, p)
        group[:G] = np.arange(1, G + 1)                      # labels 1..G all present
        y = rng.standard_normal(p) * np.exp(rng.normal(0, 1, p))
        J = int(rng.integers(1, G + 1))
        vector_k = trial % 2 == 1
        k = rng.integers(0, 6, G) if vector_k else int(rng.integers(1, 6))
        want = np.zeros(p)
        norms = np.zeros(G)
        kept = {}
        for g in range(1, G + 1):
            idx = np.flatnonzero(group == g)
            kg = int(k[g - 1]) if vector_k else k
            top = idx[np.argsort(-np.abs(y[idx]), kind="stable")[:kg]]
            kept[g] = top
            norms[g - 1] = np.sum(y[top] ** 2)
        best = np.argsort(-norms, kind="stable")[:J] + 1
        for g in best:
            want[kept[g]] = y[kept[g]]
        got = oracle.project_group_sparse(y, group, J, k)
        assert np.array_equal(got, want), (trial, p, G, J, k)


def test_choose_callback_takes_the_place_of_the_references_rng(oracle):
    """_choose! (src/utilities.jl:444-458) removes the excess of a tied projection with `sample(non_zero_idx, excess,
    replace=false)`; the restatement hands that draw to the caller.  Pinned here: the callback sees the non-zero positions in
    findall order with the reference's excess -- nonzero - (k + zkeepn) with the kept covariates NOT counted in nonzero
    (:448-451), so four tied SNPs at k = 2 with an intercept lose ONE, not two; what it returns is what goes; a draw that
    repeats the restatement's own deterministic rule (the highest index, all |b| being tied) gives the identical fit; a draw
    outside the list is an error, not a silently different model."""
    from conftest import seeded_draw, tied_case
    cols, y, tied = tied_case()
    x = oracle.Mat.from_bed_columns(cols, 1000)
    plain = oracle.fit_iht(x, y, None, k=2)
    assert plain["choose_fired"]
    log = []
    same = oracle.fit_iht(x, y, None, k=2, choose=lambda kind, lst, excess: (log.append((kind, lst.tolist(), excess)), lst[-excess:])[1])
    assert log == [(0, tied, 1)] * 2                 # at the initial support and after the first step; later steps have no tie at the cut
    np.testing.assert_array_equal(same["beta"], plain["beta"])
    assert same["iter"] == plain["iter"] and same["logl"] == plain["logl"]
    assert sorted(np.flatnonzero(plain["beta"])) == tied[:3]
    log2 = []
    drawn = oracle.fit_iht(x, y, None, k=2, choose=seeded_draw(11, log2))
    assert log2 == log and drawn["choose_fired"]
    rng = np.random.default_rng(11)
    picks = [int(rng.choice(tied, size=1, replace=False)[0]) for _ in log2]
    assert sorted(np.flatnonzero(drawn["beta"])) == sorted(set(tied) - {picks[1]})   # the SNP drawn after the first step is the one left out
    assert drawn["logl"] == pytest.approx(plain["logl"], rel=1e-12)                    # the copies are interchangeable
    with pytest.raises(RuntimeError):
        oracle.fit_iht(x, y, None, k=2, choose=lambda kind, lst, excess: np.array([5]))              # not in the list
    with pytest.raises(RuntimeError):
        oracle.fit_iht(x, y, None, k=1, choose=lambda kind, lst, excess: lst[:1].repeat(excess))     # excess = 2: the same SNP twice
    # multivariate (src/multivariate.jl:310-351): shuffle!(B_nz_idx), shuffle!(C_nz_idx), then the first `excess` go
    Y = np.vstack([y, np.random.default_rng(5).standard_normal(1000)])
    mplain = oracle.fit_mv(x, Y, None, k=1)
    assert mplain["choose_fired"]
    mlog = []
    mdrawn = oracle.fit_mv(x, Y, None, k=1, choose=seeded_draw(12, mlog))
    # B list, then C list -- unless a list is empty (intercept-only Z with zkeep: C_nz_idx is), which is not handed over:
    # shuffle! of an empty vector draws nothing from the RNG
    assert mlog and all(len(c[1]) > 0 for c in mlog) and mlog[0][0] == 1 and {c[0] for c in mlog} <= {1, 2}
    assert mlog[0] == (1, [2 * j for j in tied], 1)      # trait 0 of the four tied SNPs in eachindex order; excess = 4 - (k + r)
    rng = np.random.default_rng(12)
    first = rng.permutation(mlog[0][1])
    assert mdrawn["choose_fired"] and np.count_nonzero(mdrawn["B"]) <= 3
    assert len(mlog) == 1                                # one tied projection: the first step (the initial support is not projected through project_k!(v), multivariate.jl:436-445)
    assert sorted(np.flatnonzero(mdrawn["B"].ravel(order="F"))) == sorted(set(mlog[0][1]) - {int(first[0])})


def test_group_norms_round_the_square_then_the_sum(oracle):
    """utilities.jl:626 `group_norm[n] + y[j]^2`: square rounded, then the sum (no fma): 2.2^2 + 1.8^2 + 1.3^2 = 9.770000000000001 beats
    2.0^2 + 1.7^2 + 1.2^2 + 1.2^2 = 9.77; fused, they tie and the lower label wins (the GPU test of the same name)."""
    y = np.array([-2.0, 1.7, -1.2, 1.2, 2.2, -1.8, 1.3, 0.05])
    group = np.array([1, 1, 1, 1, 2, 2, 2, 3])
    want = np.array([0, 0, 0, 0, 2.2, -1.8, 1.3, 0])
    assert np.array_equal(oracle.project_group_sparse(y, group, 1, 4), want)
    assert np.array_equal(oracle.project_group_sparse(y, group, 1, np.array([4, 3, 1])), want)
