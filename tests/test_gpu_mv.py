"""the multivariate fit and its cross-validation against the oracle (split out of test_gpu_parity.py in round 6)."""
import json
import os

import numpy as np
import pytest

from conftest import FIX, GOLD, ROOT, SweepTally, check_recorded_cv_curve, free_device_bytes, hash_folds, make_bed, perm_folds, seeded_draw, tied_case
from gpu_helpers import _BT_TIE, _NUDGES, _config3_problem, _config4_problem, _dosages, _exact_xtv, _mv_problem, _run_probe_snippet, _same_fit, _sim, _unstable, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("r,k,q", [(2, 10, 1), (3, 12, 2), (5, 20, 1)])
def test_multivariate_fit_vs_oracle(mih, oracle, normal_pair, r, k, q):
    """fit_iht with MvNormal traits (src/multivariate.jl; test/multivariate_test.jl:84-118)."""
    x, ox = normal_pair
    rng = np.random.default_rng(40 + r)
    Y, Z = _mv_problem(oracle, ox, rng, r, k, q)
    zk = None if q == 1 else [1] + [0] * (q - 1)
    res = mih.fit_iht(Y, x, Z, k=k, zkeep=zk, verbose=False)
    o = oracle.fit_mv(ox, Y, Z, k=k, zkeep=zk)
    assert res.iter == o["iter"] and res.iter >= 5
    assert np.array_equal(res.beta != 0, o["B"] != 0)                         # bit-exact support
    np.testing.assert_allclose(res.beta, o["B"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.c, o["C"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.Σ, o["Sigma"], rtol=1e-6)
    np.testing.assert_allclose(res.σg, o["pve"], rtol=1e-6)
    assert res.logl == pytest.approx(o["logl"], rel=1e-9)
    assert list(res.trace["backtracks"]) == list(o["bt_trace"])
    assert res.beta.shape == (r, x.p) and np.count_nonzero(res.beta) <= k and np.all(res.σg > 0)

def test_multivariate_shipped_data_and_cv(mih, oracle):
    """data/multivariate.* (true Sigma shipped) + cv_iht on multivariate traits (cv_iht_test.jl:259-284)."""
    n = 1000
    bed = mih.read_bed(os.path.join(FIX, "multivariate.bed"), n)
    x = mih.SnpLinAlg(bed, n, center=True, scale=True, impute=True)
    ox = oracle.Mat.from_bed_columns(bed, n)
    Y = np.loadtxt(os.path.join(FIX, "multivariate.phen"), delimiter=",").T
    S = np.loadtxt(os.path.join(FIX, "multivariate.trait.cov"), delimiter=",")
    res = mih.fit_iht(Y, x, None, k=10, verbose=False)
    o = oracle.fit_mv(ox, Y, None, k=10)
    assert res.iter == o["iter"]
    assert np.array_equal(res.beta != 0, o["B"] != 0)
    np.testing.assert_allclose(res.beta, o["B"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.Σ, S, atol=0.12)
    with pytest.raises(mih.MendelIHTError):
        mih.fit_iht(Y.T, x, None, k=10, verbose=False)                         # un-transposed input: DimensionMismatch
    folds = hash_folds(n, 3)
    mse, raw = mih.cv_iht(Y, x, None, path=[2, 6, 10, 14], q=3, folds=folds, verbose=False, return_raw=True)
    omse, oraw = oracle.cv_mv(ox, Y, None, path=[2, 6, 10, 14], q=3, folds=folds)
    np.testing.assert_allclose(raw, oraw, rtol=1e-4)
    np.testing.assert_allclose(mse, omse, rtol=1e-4)
    assert np.all(mse > 0)

def test_multivariate_init_beta(mih, oracle, normal_pair):
    """init_beta=true for MvNormal traits (initialize_beta!(::mIHTVariable), multivariate.jl:519-558; used by
    test/multivariate.ipynb and test/NFBC-chr21.ipynb): shipped data, a covariate problem with a train mask, CV."""
    n = 1000
    bed = mih.read_bed(os.path.join(FIX, "multivariate.bed"), n)
    x = mih.SnpLinAlg(bed, n, center=True, scale=True, impute=True)
    ox = oracle.Mat.from_bed_columns(bed, n)
    Y = np.loadtxt(os.path.join(FIX, "multivariate.phen"), delimiter=",").T
    res = mih.fit_iht(Y, x, None, k=10, init_beta=True, verbose=False)
    o = oracle.fit_mv(ox, Y, None, k=10, init_beta=True)
    plain = oracle.fit_mv(ox, Y, None, k=10)
    assert res.iter == o["iter"] and list(res.trace["backtracks"]) == list(o["bt_trace"])
    assert np.array_equal(res.beta != 0, o["B"] != 0)
    np.testing.assert_allclose(res.beta, o["B"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.c, o["C"], rtol=1e-5)
    np.testing.assert_allclose(res.trace["logl"], o["logl_trace"], rtol=1e-9)
    assert o["logl"] != plain["logl"]                                          # the start really differs
    # covariates (one of them not kept), missing genotypes in X, a train mask
    x2, ox2 = normal_pair
    rng = np.random.default_rng(77)
    Y2, Z2 = _mv_problem(oracle, ox2, rng, 3, 9, 3)
    train = (np.arange(x2.n) % 4 != 1).astype(np.uint8)
    zk = [1, 1, 0]
    r2 = mih.fit_iht(Y2, x2, Z2, k=9, zkeep=zk, init_beta=True, train=train, verbose=False)
    o2 = oracle.fit_mv(ox2, Y2, Z2, k=9, zkeep=zk, init_beta=True, train=train)
    assert r2.iter == o2["iter"]
    assert np.array_equal(r2.beta != 0, o2["B"] != 0)
    np.testing.assert_allclose(r2.beta, o2["B"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(r2.c, o2["C"], rtol=1e-5, atol=1e-12)
    assert r2.logl == pytest.approx(o2["logl"], rel=1e-9)
    folds = hash_folds(n, 3)
    mse = mih.cv_iht(Y, x, None, path=[3, 8], q=3, folds=folds, init_beta=True, verbose=False)
    omse, _ = oracle.cv_mv(ox, Y, None, path=[3, 8], q=3, folds=folds, init_beta=True)
    np.testing.assert_allclose(mse, omse, rtol=1e-4)

def test_config4_multivariate_r10(mih, oracle, normal_pair):
    """configs[4]'s trait count: MvNormal with r = 10 traits (10 x 10 pivoted Cholesky step size, ten residuals in one
    fused four-operand pass with two idle residual slots) against oracle.fit_mv, plus cv_iht with r = 10 (two fits in
    flight per lock-step round)."""
    x, ox = normal_pair
    rng = np.random.default_rng(410)
    Y, Z = _mv_problem(oracle, ox, rng, 10, 40, 2)
    res = mih.fit_iht(Y, x, Z, k=40, verbose=False)
    o = oracle.fit_mv(ox, Y, Z, k=40)
    assert res.iter == o["iter"] and res.iter >= 5
    assert np.array_equal(res.beta != 0, o["B"] != 0)
    np.testing.assert_allclose(res.beta, o["B"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.c, o["C"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.Σ, o["Sigma"], rtol=1e-6)
    assert res.logl == pytest.approx(o["logl"], rel=1e-9)
    assert list(res.trace["backtracks"]) == list(o["bt_trace"])
    assert res.beta.shape == (10, x.p) and np.count_nonzero(res.beta) <= 40
    folds = hash_folds(ox.n, 3)
    path = [10, 25, 40, 60]
    mse, raw = mih.cv_iht(Y, x, Z, path=path, q=3, folds=folds, verbose=False, return_raw=True)
    omse, oraw = oracle.cv_mv(ox, Y, Z, path=path, q=3, folds=folds)
    np.testing.assert_allclose(raw, oraw, rtol=1e-6)
    np.testing.assert_allclose(mse, omse, rtol=1e-6)

@pytest.mark.parametrize("r", [6, 7, 8, 9, 11, 12])
def test_multivariate_trait_counts_of_every_product_kernel_shape(mih, oracle, normal_pair, r):
    """The multi-trait X*B kernel is instantiated for 4, 6, 8, 10 and 12 traits per thread (csrc/xv.hip, k_xv_snp_cached_mt:
    padded coefficient records and column offsets, batches of eight columns): trait counts the sweeps (2 .. 5) and configs[4]
    (10) do not reach, with support sizes that are not multiples of eight, against oracle.fit_mv."""
    x, ox = normal_pair
    rng = np.random.default_rng(600 + r)
    k = 13 + r                                        # 19 .. 25 entries: supports of 8 m + 1 .. 8 m + 7 columns among them
    Y, Z = _mv_problem(oracle, ox, rng, r, 9, 2)
    res = mih.fit_iht(Y, x, Z, k=k, verbose=False, max_iter=30)
    o = oracle.fit_mv(ox, Y, Z, k=k, max_iter=30)
    assert res.iter == o["iter"] and res.iter >= 4
    assert np.array_equal(res.beta != 0, o["B"] != 0)
    np.testing.assert_allclose(res.beta, o["B"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.c, o["C"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.Σ, o["Sigma"], rtol=1e-6)
    assert res.logl == pytest.approx(o["logl"], rel=1e-9)
    assert list(res.trace["backtracks"]) == list(o["bt_trace"])

def test_config4_multivariate_r10_k500_against_oracle(mih, oracle):
    """BASELINE configs[4]'s model size: MvNormal with r = 10 traits and k = 500 non-zero entries at
    p = 20 000 SNPs against oracle.fit_mv (multivariate.jl:99-127: top-k over all r * p entries): same iterations, backtracks
    and support, B and C to 1e-5, Sigma and the loglikelihood."""
    n, p, r, k = 3_000, 20_000, 10, 500
    x = mih.SnpLinAlg.synthetic(n, p, seed=41)
    ox = oracle.Mat.from_bed_columns(x.export_bed(), n)
    rng = np.random.default_rng(4100)
    Y, Z = _mv_problem(oracle, ox, rng, r, k, 2)
    res = mih.fit_iht(Y, x, Z, k=k, verbose=False, max_iter=60)
    o = oracle.fit_mv(ox, Y, Z, k=k, max_iter=60)
    assert res.iter == o["iter"] and res.iter >= 5
    assert list(res.trace["backtracks"]) == list(o["bt_trace"])
    assert np.count_nonzero(res.beta) == np.count_nonzero(o["B"]) and 400 <= np.count_nonzero(res.beta) <= k
    assert np.array_equal(res.beta != 0, o["B"] != 0)
    np.testing.assert_allclose(res.beta, o["B"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.c, o["C"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(res.Σ, o["Sigma"], rtol=1e-6)
    assert res.logl == pytest.approx(o["logl"], rel=1e-9)


def test_multivariate_fit_with_a_gross_outlier_in_one_trait(mih, oracle, normal_pair):
    """(round 6) The rows of T1 = Gamma * resid are the residuals of the multivariate score (multivariate.jl:84-86).  ONE sample with
    a trait value 4e7 x the rest: the fit keeps the oracle's iteration log and support, B / C / Sigma to the Gaussian tolerance and
    its loglikelihood trace to 1e-10.  (The outlier guard of csrc/peel.h fires once, for the initial score: from then on
    Gamma = inv(R R' / n) scales the outlying trait down by its own variance -- Gamma_11 ~ n / 1.6e15 -- and no row of T1 towers.)"""
    x, ox = normal_pair
    rng = np.random.default_rng(515)
    r, k = 3, 9
    Y, Z = _mv_problem(oracle, ox, rng, r, k, 1)
    Y[1, 123] = 4.0e7
    mih.profile_enable(x, True)
    mih.profile_counters(x, reset=True)
    res = mih.fit_iht(Y, x, Z, k=k, verbose=False)
    cnt = mih.profile_counters(x, reset=True)
    mih.profile_enable(x, False)
    o = oracle.fit_mv(ox, Y, Z, k=k)
    assert res.iter == o["iter"] and list(res.trace["backtracks"]) == list(o["bt_trace"])
    assert np.array_equal(res.beta != 0, o["B"] != 0)
    np.testing.assert_allclose(res.beta, o["B"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(res.c, o["C"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(res.Σ, o["Sigma"], rtol=1e-6)
    np.testing.assert_allclose(res.trace["logl"], o["logl_trace"], rtol=1e-10)
    assert 1 <= cnt["peeled_residuals"] <= r * res.iter, cnt
