"""The classes of ill-conditioned input the randomized GPU sweeps set aside, pinned on the CPU: every seed of
tools/fuzz_parity.py that ended red on an input no floating-point implementation can be held to is replayed here against the ORACLE
ALONE -- the case generators of tests/test_gpu_parity.py with a stand-in for the device matrix -- and the diagnostic the sweeps'
rule rests on must fire: the oracle's own report (orc_result.bt_cond / ib_cond) or its disagreement with itself under
ulp-sized nudges of the input.  No GPU, no device library: the oracle is the subject here."""
import numpy as np
import pytest

import mendeliht_amd as _real
import test_gpu_parity as T


class _NoDevice:                                   # the generators build a device matrix beside the oracle's; nothing here uses it
    def __init__(self, *a, **k):
        pass


class _Mirror:                                     # the mirror's distribution / link classes are plain Python; only the matrices need a device
    SnpLinAlg = _NoDevice
    DenseMatrix = _NoDevice

    def __getattr__(self, name):
        return getattr(_real, name)


FAKE = _Mirror()
FAMS3 = [("normal", "identity", FAKE.Normal, FAKE.IdentityLink, 1e-5), ("bernoulli", "logit", FAKE.Bernoulli, FAKE.LogitLink, 1e-4),
         ("poisson", "log", FAKE.Poisson, FAKE.LogLink, 1e-4)]
FAMS4 = [f[:4] for f in FAMS3] + [("negbin", "log", FAKE.NegativeBinomial, FAKE.LogLink)]


def _replay(gen, seed, trial, *extra):
    rng = np.random.default_rng(seed)
    for t in range(trial + 1):
        case = gen(FAKE, _ORACLE[0], rng, t, *extra)
    return case


_ORACLE = [None]


@pytest.fixture(autouse=True)
def _bind(oracle):
    _ORACLE[0] = oracle


def test_backtracking_decision_between_equal_loglikelihoods(oracle):
    """Seed 9878 (first sweep, trial 8): the converging step's loglikelihood equals the previous one to the last bits; the oracle
    halves the step twice, reports the margin of that decision (bt_cond), and is perfectly stable under nudges -- which is why the
    sweep needs the diagnostic."""
    n, p, k, miss, q, od, ol, D, L, tol, x, ox, y, z, kw = _replay(T._fits_case, 9878, 8, FAMS3)
    assert (n, p, k, od, miss) == (2142, 186, 2, "normal", "dense64") and sorted(kw) == ["train", "weight"]
    o = oracle.fit_iht(ox, y, z, k=k, dist=od, link=ol, max_iter=60, **kw)
    assert o["iter"] == 5 and list(o["bt_trace"]) == [0, 0, 0, 0, 2]
    assert o["bt_cond"] < T._BT_TIE
    for g in T._NUDGES[:3]:                        # the nudges do not see it: same iterations, estimates equal to 1e-15
        o2 = oracle.fit_iht(ox, y, z * g, k=k, dist=od, link=ol, max_iter=60, **kw)
        assert o2["iter"] == 5 and np.abs(o2["beta"] - o["beta"]).max() < 1e-15
    plain = oracle.fit_iht(ox, y, z, k=k, dist=od, link=ol, max_iter=4, **kw)      # a fit that never compares two close loglikelihoods
    assert plain["bt_cond"] > 1000 * T._BT_TIE      # (its closest decision: a relative gain of 2.6e-10, far from a tie)


def test_negbin_r_running_off(oracle):
    """Seeds 10168 (model paths, trial 4, k = 12) and 10864 (trial 1, k = 8, with debias): counts without overdispersion, r runs off
    to 1e7 and beyond and differs by orders of magnitude from one ulp-sized nudge to the next; with debias the fit of the original
    input even ends in the reference's error while every nudged one finishes."""
    n, p, q, od, ol, D, L, x, ox, y, z, path, kw, okw, d = _replay(T._path_case, 10168, 4, FAMS4)
    assert od == "negbin" and path == [5, 6, 12] and okw.get("est_r") == "newton"
    rs = [oracle.fit_iht(ox, y, z * g, k=12, dist=od, link=ol, max_iter=100, **okw)["nb_r"] for g in [1.0] + T._NUDGES]
    assert min(rs) > 1e6 and max(rs) / min(rs) > 3.0
    small = oracle.fit_iht(ox, y, z, k=5, dist=od, link=ol, max_iter=100, **okw)     # the same data at k = 5: r settles (13.29), the sweep compares it
    assert small["nb_r"] < 100.0
    n, p, q, od, ol, D, L, x, ox, y, z, path, kw, okw, d = _replay(T._path_case, 10864, 1, FAMS4)
    assert od == "negbin" and path == [2, 7, 8] and okw.get("debias")
    with pytest.raises(RuntimeError):
        oracle.fit_iht(ox, y, z, k=8, dist=od, link=ol, max_iter=100, **okw)
    for g in T._NUDGES:
        o = oracle.fit_iht(ox, y, z * g, k=8, dist=od, link=ol, max_iter=100, **okw)
        assert o["iter"] == 100 and o["nb_r"] > 1e6


def test_init_beta_with_a_predictor_constant_over_the_training_rows(oracle):
    """Seeds 10545 (multivariate fits, trial 6) and 10137 (multivariate CV, trial 0, fold 1): a SNP monomorphic in the training rows
    makes the second pivot of linreg!'s Cholesky a rounding residue; the univariate oracle on those rows reports it (ib_cond)."""
    n, p, r, q, k, miss, x, ox, Y, Z, kw = _replay(T._mvfit_case, 10545, 6)
    assert (n, p, r) == (193, 392, 4) and kw.get("init_beta")
    one = oracle.fit_iht(ox, Y[0], None, k=1, max_iter=1, train=kw.get("train"), init_beta=True)
    assert one["ib_cond"] < 1e-10
    n, p, r, qz, q, x, ox, Y, Z, path, folds, extra = _replay(T._mvcv_case, 10137, 0)
    assert (n, p, r, q) == (201, 173, 4, 2) and extra.get("init_beta")
    conds = [oracle.fit_iht(ox, Y[0], None, k=1, max_iter=1, train=(folds != f + 1).astype(np.uint8), init_beta=True)["ib_cond"] for f in range(q)]
    assert conds[0] < 1e-10 and conds[1] > 0.1      # fold 1's training rows hold the constant SNP, fold 2's do not


def test_zero_over_zero_step_size(oracle):
    """Seed 10328 (keyword sweep, trial 6: Gamma / log with debias, k = 1): after debias! has refitted the one-SNP support the score on
    it is a rounding residue -- the share of the squared score on the support (eta_cond) is 5e-19 -- and iht_stepsize! divides two such
    residues; the loglikelihoods of the step that follows tie exactly (bt_cond = 0)."""
    rng = np.random.default_rng(10328)
    for t in range(7):
        x, ox, y, z, k, kw, okw, both, tol, fam, tag = T._options_case(FAKE, oracle, rng, t)
    assert tag[1:3] == (209, 280) and fam == "gamma" and both.get("debias")
    o = oracle.fit_iht(ox, y, z, k=k, max_iter=40, **okw, **both)
    assert o["eta_cond"] < 1e-18 and o["bt_cond"] < T._BT_TIE
    rng = np.random.default_rng(10328)
    for t in range(2):
        x, ox, y, z, k, kw, okw, both, tol, fam, tag = T._options_case(FAKE, oracle, rng, t)   # an ordinary trial of the same seed (NegBin, :MM): neither fires
    o = oracle.fit_iht(ox, y, z, k=k, max_iter=40, **okw, **both)
    assert o["eta_cond"] > 1e-12 and o["bt_cond"] > 1000 * T._BT_TIE


def test_debias_error_that_comes_and_goes_under_nudges_in_a_model_path(oracle):
    """Seed 16136 (round 6's campaign: model paths, trial 2, NegBin / log with est_r = :Newton and debias, n = 151, k = 11): the HIP
    path ended in the reference's "debias: step-halving failed" error where the oracle's fit of the original input finishes -- and
    the oracle itself ends in that error under the first ulp-sized nudge of z, its loglikelihood moving in the fifth digit from nudge
    to nudge (class (d) of DESIGN 4, here in the model-path sweep, whose harness compares loglikelihoods and does not catch the
    library's error).  The smaller model sizes of the same path are stable and are what the sweep checks."""
    n, p, q, od, ol, D, L, x, ox, y, z, path, kw, okw, d = _replay(T._path_case, 16136, 2, FAMS4)
    assert (n, p, od) == (151, 268, "negbin") and path == [5, 6, 7, 8, 10, 11] and okw.get("debias") and okw.get("est_r") == "newton"
    plain = oracle.fit_iht(ox, y, z, k=11, dist=od, link=ol, max_iter=100, **okw)
    outcomes = []
    for g in T._NUDGES:
        try:
            outcomes.append(oracle.fit_iht(ox, y, z * g, k=11, dist=od, link=ol, max_iter=100, **okw)["logl"])
        except RuntimeError:
            outcomes.append(None)
    assert any(v is None for v in outcomes)                                   # the reference's error, under a nudge
    done = [v for v in outcomes if v is not None] + [plain["logl"]]
    assert max(done) - min(done) > 1e-6 * abs(plain["logl"])                   # ... and where it finishes, a different optimum each time
    stable = [oracle.fit_iht(ox, y, z * g, k=5, dist=od, link=ol, max_iter=100, **okw)["logl"] for g in [1.0] + T._NUDGES[:3]]
    assert max(stable) - min(stable) < 1e-9 * abs(stable[0])


def _options_oracle(oracle, seed, trial):
    x, ox, y, z, k, kw, okw, both, tol, fam, tag = _replay(T._options_case, seed, trial)

    def run(m, yy, zz):
        try:
            return oracle.fit_iht(m, yy, zz, k=k, max_iter=40, **okw, **both), None
        except RuntimeError as e:
            return None, e
    return ox, y, z, tol, fam, both, run


def test_row_orders_reach_what_the_nudges_of_z_cannot(oracle):
    """Round 6 ("differs after a step that used up max_step backtracks" was accepted by argument).  Seed
    16133, trial 11: Poisson with the sqrt link and debias.  Scaling z by an ulp leaves the oracle's answer where it is -- debias!
    fits the SNP columns alone -- but the same problem with its samples in another order (every sum over the samples re-associated)
    moves it beyond the tolerance: the probe the sweeps now apply before they hold the device to such a trajectory."""
    ox, y, z, tol, fam, both, run = _options_oracle(oracle, 16133, 11)
    assert fam == "poisson_sqrt" and both.get("debias")
    o, _ = run(ox, y, z)
    strip = lambda d: {key: d[key] for key in ("iter", "beta", "c", "logl")}
    for g in T._NUDGES:
        v, _ = run(ox, y, z * g)
        assert not T._unstable(strip(o), dict(strip(v), c=v["c"] * g), tol, atol=1e-9)
    moved = []
    for pm in T._row_orders(len(y)):
        v, _ = run(T._rows_permuted(oracle, ox, pm), y[pm], z[pm])
        moved.append(v is None or T._unstable(strip(o), strip(v), tol, atol=1e-9))
    assert any(moved), moved


def test_row_order_leaves_a_well_conditioned_fit_alone(oracle):
    """... and the probe is not a blanket excuse: the reference's recorded-size Normal fit of the first sweep (seed 20260, trial 0) is
    the same fit under every row order, to 1e-9 of its estimates."""
    n, p, k, miss, q, od, ol, D, L, tol, x, ox, y, z, kw = _replay(T._fits_case, 20260, 0, FAMS3)
    o = oracle.fit_iht(ox, y, z, k=k, dist=od, link=ol, max_iter=60, **kw)
    for pm in T._row_orders(len(y)):
        kr = dict(kw, train=kw["train"][pm]) if "train" in kw else kw
        v = oracle.fit_iht(T._rows_permuted(oracle, ox, pm), y[pm], z[pm], k=k, dist=od, link=ol, max_iter=60, **kr)
        assert v["iter"] == o["iter"] and np.array_equal(np.flatnonzero(v["beta"]), np.flatnonzero(o["beta"]))
        np.testing.assert_allclose(v["beta"], o["beta"], rtol=1e-9, atol=1e-12)
        assert v["logl"] == pytest.approx(o["logl"], rel=1e-12)


def test_debias_refit_that_crawls_by_halved_steps(oracle):
    """Seed 16330, trial 0 (Poisson / sqrt, debias, k = 8): at iteration 5 the oracle's debias! refit runs out of its 30 IRLS iterations
    -- under EVERY row order -- after halving its steps down to 1/256; the device's 28th step, halved twice, lowers the deviance
    by less than the tolerance and counts as converged (profiles/README.md, round 6).  The oracle reports the crawl
    (orc_result.db_minstep, also on the error path); a refit that never halves reports 1."""
    ox, y, z, tol, fam, both, run = _options_oracle(oracle, 16330, 0)
    o, e = run(ox, y, z)
    assert o is None and e.db_minstep < 0.05
    for pm in T._row_orders(len(y), 2):
        v, e2 = run(T._rows_permuted(oracle, ox, pm), y[pm], z[pm])
        assert v is None and e2.db_minstep < 0.05
    n, p, k, miss, q, od, ol, D, L, tol, x, ox, y, z, kw = _replay(T._fits_case, 20260, 0, FAMS3)
    assert oracle.fit_iht(ox, y, z, k=k, dist=od, link=ol, max_iter=60, debias=True, **kw)["db_minstep"] == 1.0


def test_oracle_spread_that_takes_up_most_of_the_tolerance(oracle):
    """Seed 16276, trial 2 (Poisson / sqrt, debias): the oracle reproduces itself under re-association to within the tolerance -- but
    an effect of 0.008 moves by up to 8e-7 (1e-4 of it is 8e-7): the sweeps hold the device to the tolerance plus four times that
    spread (_within_own_spread), and to nothing looser."""
    ox, y, z, tol, fam, both, run = _options_oracle(oracle, 16276, 2)
    o, _ = run(ox, y, z)
    strip = lambda d: {key: d[key] for key in ("iter", "beta", "c", "logl")}
    rows = [run(T._rows_permuted(oracle, ox, pm), y[pm], z[pm])[0] for pm in T._row_orders(len(y))]
    assert all(v is not None and not T._unstable(strip(o), strip(v), tol, atol=1e-9) for v in rows)
    supp = np.flatnonzero(o["beta"])
    spread = np.max([np.abs(v["beta"][supp] - o["beta"][supp]) for v in rows], axis=0)
    assert 0.2 * tol < np.max(spread / np.abs(o["beta"][supp])) < tol
    tols = dict(beta=(tol, 1e-9), c=(tol, 1e-9), logl=(1e-7, 0.0))
    near = dict(beta=o["beta"] * (1 + 1.9 * tol * (np.abs(o["beta"]) == np.abs(o["beta"][supp]).min())), c=o["c"], logl=o["logl"])       # the device's answer was 1.84e-4 off on the smallest effect
    far = dict(near, beta=o["beta"] * (1 + 20 * tol))
    assert T._within_own_spread(o, rows, near, tols) and not T._within_own_spread(o, rows, far, tols)
    assert not T._within_own_spread(o, [o], near, tols)            # (no spread, no allowance)


def test_recorded_run_under_every_row_order(oracle, normal_data):
    """The probe against the reference's own record (docs/src/man/examples.md:230-267, tests/golden/golden_normal_k7.json): the recorded
    fit with its 1000 samples in four other orders is the recorded fit -- iterations, backtracks, support, the loglikelihood of
    every iteration to 1e-11.  What a row order moves on a well-conditioned problem is rounding, nothing else."""
    import json, os
    g = json.load(open(os.path.join(T.GOLD, "golden_normal_k7.json")))
    x = oracle.Mat.from_bed_file(normal_data["bed"], normal_data["n"])
    y, z = normal_data["y"], normal_data["z"]
    for pm in T._row_orders(normal_data["n"]):
        r = oracle.fit_iht(T._rows_permuted(oracle, x, pm), y[pm], z[pm], k=7)
        assert r["iter"] == g["iterations"] and list(r["bt_trace"]) == g["backtracks"]
        assert list(np.flatnonzero(r["beta"]) + 1) == g["positions_1based"]
        np.testing.assert_allclose(r["logl_trace"], g["logl"], rtol=1e-11)
        np.testing.assert_allclose(r["c"], g["c_printed"], rtol=5e-6)
