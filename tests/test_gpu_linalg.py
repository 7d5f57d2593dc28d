"""SURVEY 8 rows a1 (X'r over the 2-bit matrix), a7 / a8 (projections), f2 (ingest, naive_impute) on the GPU against the oracle,
exact rational arithmetic and the round-1 kernel families (split out of test_gpu_parity.py in round 6)."""
import json
import os

import numpy as np
import pytest

from conftest import FIX, GOLD, ROOT, SweepTally, check_recorded_cv_curve, free_device_bytes, hash_folds, make_bed, perm_folds, seeded_draw, tied_case
from gpu_helpers import peel_rule, _BT_TIE, _NUDGES, _config3_problem, _config4_problem, _dosages, _exact_xtv, _mv_problem, _run_probe_snippet, _same_fit, _sim, _unstable, rel

pytestmark = pytest.mark.gpu


_ROUND1_SNIPPET = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import mendeliht_amd as m
assert m.using_probes()
n = 1000
x = m.SnpLinAlg(m.read_bed(os.path.join(sys.argv[1], "tests", "fixtures", "normal.bed"), n), n, center=True, scale=True, impute=True)
r = np.random.default_rng(0).standard_normal(n)
out = {"default": x.xtv(r), "base1316": x.xtv(r, xtv_digits=1316)}
nv = 0
while True:                                                  # round 1's per-wave-load shapes
    try:
        m.probe_set(variant=nv)
    except m.MendelIHTError:
        break
    out[f"variant{nv}"] = x.xtv(r, xtv_digits=1316)
    nv += 1
m.probe_set(variant=-1)
for mv in (9, 10, 11, 12, 13, 14):                           # the register-staged single-operand shapes
    m.probe_set(multi_variant=mv)
    out[f"multi{mv}"] = x.xtv(r, xtv_digits=1316)
m.probe_set(multi_variant=0)
R = np.asfortranarray(np.random.default_rng(1).standard_normal((n, 7)))
out["R7_default"] = x.xtv(R)
m.probe_set(multi_variant=6)                                 # round 1's register-staged FP6 kernels (32x32x64)
out["R7_regstaged"] = x.xtv(R)
m.probe_set(multi_variant=20)                                # the 32x32x64 LDS-DMA ring
out["R7_ring32"] = x.xtv(R)
m.probe_set(multi_variant=0)
xs = m.SnpLinAlg.synthetic(500_000, 64, seed=2024)           # full row count (eight row slices)
r1 = np.random.default_rng(5).standard_normal(500_000)
out["big_default"] = xs.xtv(r1)
nb = 0
while True:
    try:
        m.probe_set(variant=nb)
    except m.MendelIHTError:
        break
    out[f"big_variant{nb}"] = xs.xtv(r1)
    nb += 1
m.probe_set(variant=-1)
np.savez(sys.argv[2], **out)
"""

_TAIL_COLUMNS_SNIPPET = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import mendeliht_amd as m
from mendeliht_amd import api
assert m.using_probes()
n, mcap = 1000, 19
x = m.SnpLinAlg(m.read_bed(os.path.join(sys.argv[1], "tests", "fixtures", "normal.bed"), n), n, center=True, scale=True, impute=True)
rng = np.random.default_rng(7)
R = np.asfortranarray(rng.standard_normal((n, mcap)) * np.exp(rng.uniform(-3, 3, mcap)))      # a different scale per residual
ms = np.array([19] + list(range(1, 19)) + [19], dtype=np.int32)       # a full pass first, then every shorter count, then full again
out = np.zeros(int(ms.sum()) * x.p)
api._check(api.lib().mih_probe_xtv_sequence(x._h, api._p(R), mcap, api._p(ms), ms.size, 0, api._p(out)))
np.savez(sys.argv[2], R=R, ms=ms, out=out)
"""

def test_flat_packing_ignores_what_an_earlier_pass_left_in_the_tail_columns(mih, normal_pair, tmp_path):
    """ With the flat digit packing k_digits writes only the columns of the call's m residuals; the unused tail columns
    of the last operand keep the digits of an earlier call.  A lock-step lane does exactly that from round to round.  Here: 19
    residuals, then 1 .. 18, then 19 again on ONE workspace (measurement build: mih_probe_xtv_sequence) -- every result must be
    bit for bit what a FRESH workspace gives for the same residuals (mih_xtv_batched_fmt, product library, this process)."""
    x, _ = normal_pair
    got = _run_probe_snippet(_TAIL_COLUMNS_SNIPPET, tmp_path / "tail.npz")
    R, ms, out = got["R"], got["ms"], got["out"]
    fresh = {int(mm): x.xtv(np.asfortranarray(R[:, :mm])).reshape(x.p, -1, order="F") for mm in sorted(set(ms.tolist()))}
    off = 0
    for mm in ms.tolist():
        blk = out[off:off + mm * x.p].reshape(x.p, mm, order="F")
        assert np.array_equal(blk, fresh[mm]), mm
        off += mm * x.p
    assert np.array_equal(fresh[19][:, :7], fresh[7])              # and a residual's X'r does not depend on the company it rides with

def test_mu_sinv_and_xtv_against_oracle(mih, normal_pair):
    x, ox = normal_pair
    assert not mih.using_probes()                                   # the tests run on the product library
    mu, s = x.mu_sigma()
    omu, os_ = ox.mu_sinv()
    assert np.array_equal(mu, omu) and np.array_equal(s, os_)
    r = np.random.default_rng(0).standard_normal(x.n)
    ref = ox.xtv(r)
    default = x.xtv(r)                                              # library default: FP6 digit planes through the LDS-DMA ring
    assert rel(default, ref) < 1e-11
    base = x.xtv(r, xtv_digits=1316)
    assert rel(base, default) < 1e-13
    assert np.array_equal(x.xtv(r), default)

def test_product_kernels_equal_the_round1_kernel_families(mih, normal_pair, oracle, tmp_path):
    """The product library has one kernel per (format family, operand count).  The measurement build (same sources,
    -DMIH_PROBES) still carries round 1's kernel families -- per-wave digit loads, digit planes staged through registers -- and
    the 32x32x64 ring: with the same row slicing they must give the product's bits, on the reference's shipped data and at the
    full row count of the benchmark (against the oracle there)."""
    x, ox = normal_pair
    got = _run_probe_snippet(_ROUND1_SNIPPET, tmp_path / "round1.npz")
    r = np.random.default_rng(0).standard_normal(x.n)
    default, base = x.xtv(r), x.xtv(r, xtv_digits=1316)            # this process: the product library
    assert np.array_equal(got["default"], default) and np.array_equal(got["base1316"], base)
    nv = sum(1 for k in got.files if k.startswith("variant"))
    assert nv >= 3
    same_slices = (5, 6, 7)                                         # shapes with one row slice, like the default at n = 1000
    for v in range(nv):
        assert rel(got[f"variant{v}"], base) < 1e-13, v             # the slice partials are rounded f64 sums of exact digit sums
        if v in same_slices:
            assert np.array_equal(got[f"variant{v}"], base), v      # same slicing: every kernel shape agrees bit for bit
    for mv in (9, 10, 11, 12, 13, 14):
        assert np.array_equal(got[f"multi{mv}"], base), mv
    R = np.asfortranarray(np.random.default_rng(1).standard_normal((x.n, 7)))
    mine = x.xtv(R)
    for k in ("R7_default", "R7_regstaged", "R7_ring32"):
        assert np.array_equal(got[k], mine), k
    xs = mih.SnpLinAlg.synthetic(500_000, 64, seed=2024)
    r1 = np.random.default_rng(5).standard_normal(500_000)
    oxs = oracle.Mat.from_bed_columns(xs.export_bed(), 500_000)
    want = oxs.xtv(r1)
    assert np.array_equal(got["big_default"], xs.xtv(r1))
    nb = sum(1 for k in got.files if k.startswith("big_variant"))
    assert nb >= 3
    for v in range(nb):
        assert rel(got[f"big_variant{v}"], want) < 1e-10, v

@pytest.mark.parametrize("n,p,miss", [(1003, 257, 0.02), (77, 33, 0.1), (5000, 100, 0.0), (2049, 64, 0.05),
                                      (16, 5, 0.0), (1, 3, 0.0), (4097, 9, 0.3)])
@pytest.mark.parametrize("flags", [(1, 1, 1), (1, 1, 0), (0, 0, 1), (1, 0, 1)])
def test_ragged_missing_flags(mih, oracle, n, p, miss, flags):
    """n not divisible by 4/16/64/1024, missing genotypes, monomorphic columns, every flag combination."""
    rng = np.random.default_rng(n * 31 + p)
    cols = make_bed(rng, n, p, miss)
    cols[0, :] = 0                                     # all-zero column: sinv = 1
    c, s, i = flags
    x = mih.SnpLinAlg(cols, n, center=c, scale=s, impute=i)
    ox = oracle.Mat.from_bed_columns(cols, n, center=c, scale=s, impute=i)
    mu, sv = x.mu_sigma()
    omu, osv = ox.mu_sinv()
    np.testing.assert_allclose(mu, omu, rtol=1e-15)
    np.testing.assert_allclose(sv, osv, rtol=1e-15)
    r = rng.standard_normal(n)
    ref = ox.xtv(r)
    # absolute error against the scale of the terms being summed (a column can sum to exactly 0)
    assert np.max(np.abs(x.xtv(r) - ref)) < 1e-11 * max(np.max(np.abs(ref)), np.sum(np.abs(r)))
    idx = np.sort(rng.choice(p, size=min(4, p), replace=False))
    val = rng.standard_normal(idx.size)
    mask = np.zeros(p, np.uint8)
    mask[idx] = 1
    coef = np.zeros(p)
    coef[idx] = val
    assert rel(x.xv_sparse(idx, val), ox.xv_masked(mask, coef)) < 1e-11
    assert np.array_equal(x.export_bed(), cols)        # encode -> device layout -> decode round trip

def test_xtv_batched_and_empty_support(mih, oracle, normal_pair):
    x, ox = normal_pair
    R = np.random.default_rng(1).standard_normal((x.n, 9))
    assert rel(x.xtv(R[:, :3]), ox.xtv_multi(R[:, :3])) < 1e-11
    singles = np.column_stack([x.xtv(R[:, v]) for v in range(9)])
    for m_rhs in range(1, 10):          # every split into 4-/2-/1-RHS passes (3 left over ride a padded 4-pass)
        assert np.array_equal(x.xtv(R[:, :m_rhs]), singles[:, :m_rhs]), m_rhs
    assert np.all(x.xv_sparse(np.zeros(0, np.int64), np.zeros(0)) == 0.0)
    with pytest.raises(mih.MendelIHTError):
        x.xv_sparse(np.array([x.p]), np.array([1.0]))

def test_synthetic_matrix_matches_oracle_after_export(mih, oracle):
    for miss in (0.0, 0.03):
        x = mih.SnpLinAlg.synthetic(3001, 130, seed=11, missing_rate=miss)
        cols = x.export_bed()
        ox = oracle.Mat.from_bed_columns(cols, 3001)
        mu, _ = x.mu_sigma()
        assert np.all((mu > 0) & (mu < 1.2))
        r = np.random.default_rng(2).standard_normal(3001)
        assert rel(x.xtv(r), ox.xtv(r)) < 1e-11
        x2 = mih.SnpLinAlg.synthetic(3001, 40, seed=11, missing_rate=miss)   # same seed: same leading columns
        assert np.array_equal(x2.export_bed(), cols[:40])

def test_project_k_device(mih, oracle):
    """project_k! (utilities.jl:553-559) incl. the reference's top-k property test (utilities_test.jl:166-176)."""
    rng = np.random.default_rng(3)
    x = rng.random(100000)
    out = mih.project_k(x, 100)
    assert np.array_equal(out, oracle.project_k(x, 100))
    assert np.count_nonzero(out) == 100
    v = rng.standard_normal(1000003)
    for k in (1, 7, 200, 5000, v.size):
        assert np.array_equal(mih.project_k(v, k), oracle.project_k(v, k)), k
    t = np.array([1.0, -2.0, 2.0, 0.5, np.inf, -0.0])
    assert np.array_equal(mih.project_k(t, 2), oracle.project_k(t, 2))     # tie at the threshold kept; Inf survives
    assert list(mih.project_k(np.array([1.0, -2.0, 2.0, 0.5]), 1)) == [0.0, -2.0, 2.0, 0.0]
    with pytest.raises(mih.MendelIHTError):
        mih.project_k(t, -1)
    with pytest.raises(mih.MendelIHTError):
        mih.project_k(t, 0)

def test_project_group_sparse_device(mih, oracle):
    """project_group_sparse! (utilities.jl:613-679) incl. the reference's property tests (utilities_test.jl:180-213)."""
    rng = np.random.default_rng(30)
    m, n, k, J = 5, 50, 3, 2
    y = rng.standard_normal(n)
    group = np.repeat(np.arange(1, m + 1), n // m)
    out = mih.project_group_sparse(y, group, J, k)
    assert np.array_equal(out, oracle.project_group_sparse(y, group, J, k))
    nzg = [np.count_nonzero(out[group == g]) for g in range(1, m + 1)]
    assert sum(c > 0 for c in nzg) == J and all(c in (0, k) for c in nzg)
    one = np.ones(n, dtype=np.int64)
    assert np.array_equal(mih.project_group_sparse(y, one, 1, 7), mih.project_k(y, 7))   # J=1 group == project_k!
    ks = np.array([1, 2, 3, 4, 5])
    assert np.array_equal(mih.project_group_sparse(y, group, 5, ks), oracle.project_group_sparse(y, group, 5, ks))
    # large, unordered labels, empty groups, ties
    p = 200003
    v = rng.standard_normal(p)
    v[::7] = np.round(v[::7], 1)                     # many exact ties
    g = rng.integers(1, 5000, size=p)
    g[g == 17] = 18                                  # an empty group
    for (JJ, kk) in [(10, 3), (4999, 1), (1, 50)]:
        assert np.array_equal(mih.project_group_sparse(v, g, JJ, kk), oracle.project_group_sparse(v, g, JJ, kk)), (JJ, kk)
    kv = rng.integers(0, 4, size=4999)
    assert np.array_equal(mih.project_group_sparse(v, g, 300, kv), oracle.project_group_sparse(v, g, 300, kv))

@pytest.mark.parametrize("mode,quantum", [(4908, 1e-12), (1308, 2e-7)])
def test_fast_digit_mode(mih, oracle, normal_pair, normal_data, mode, quantum):
    """xtv_digits = 4908: 43-bit fixed-point residuals as 8 base-49 FP6 digits, four per MFMA B operand (the
    opt-in mode for fused multi-RHS passes), and (1308): 27-bit residuals as 8 base-13 FP4 digits.  X'r stays within the format's
    quantum of the exact mode, is independent of how the residuals are grouped into passes, and fits /
    cross-validation stay inside the north_star tolerance."""
    x, ox = normal_pair
    n = x.n
    R = np.random.default_rng(5).standard_normal((n, 9))
    exact = x.xtv(R)
    mih.set_xtv_digits(mode)
    try:
        fast = x.xtv(R)
        scale = np.sqrt(n) * np.abs(R).max()                      # size of a null-SNP score
        assert np.max(np.abs(fast - exact)) < quantum * scale
        assert not np.array_equal(fast, exact)                    # it really is the other arithmetic
        singles = np.column_stack([x.xtv(R[:, v]) for v in range(9)])
        for m_rhs in range(1, 10):                                # pairs, padded 4-operand passes, odd tails
            assert np.array_equal(x.xtv(R[:, :m_rhs]), singles[:, :m_rhs]), m_rhs
        assert np.array_equal(x.xtv(R), fast)                     # reproducible
        y, z = normal_data["y"], normal_data["z"]
        res = mih.fit_iht(y, x, z, k=7, verbose=False)
        o = oracle.fit_iht(ox, y, z, k=7)
        assert res.iter == o["iter"] and np.array_equal(np.flatnonzero(res.beta), np.flatnonzero(o["beta"]))
        np.testing.assert_allclose(res.beta[res.beta != 0], o["beta"][o["beta"] != 0], rtol=1e-5)
        assert res.logl == pytest.approx(o["logl"], rel=1e-8)
        folds = hash_folds(n, 3)
        path = list(range(1, 8))
        mse = mih.cv_iht(y, x, z, path=path, q=3, folds=folds, verbose=False)
        omse, _ = oracle.cv_iht(ox, y, z, path=path, q=3, folds=folds)
        np.testing.assert_allclose(mse, omse, rtol=1e-5)
        rng = np.random.default_rng(41)
        Y, Z = _mv_problem(oracle, ox, rng, 3, 8, 2)
        rm = mih.fit_iht(Y, x, Z, k=8, verbose=False)
        om = oracle.fit_mv(ox, Y, Z, k=8)
        assert np.array_equal(rm.beta != 0, om["B"] != 0)
        np.testing.assert_allclose(rm.beta, om["B"], rtol=1e-5, atol=1e-12)
    finally:
        mih.set_xtv_digits(0)
    assert np.array_equal(x.xtv(R), exact)                        # back to the exact mode
    with pytest.raises(mih.MendelIHTError):
        mih.set_xtv_digits(20)

def test_digit_modes_agree(mih, oracle, normal_pair):
    """Every fixed-point format of the residual (xtv_digits) against the oracle's f64 X'r: the default
    (10 base-49 FP6 digits, three residuals per operand), 16 base-13 FP4 digits (two per operand) and 28 base-4
    digits (one per operand) agree to f64 rounding; the 43-bit and 27-bit formats to their quantum; each is
    independent of how residuals share operands."""
    x, ox = normal_pair
    n = x.n
    rng = np.random.default_rng(77)
    R = rng.standard_normal((n, 18)) * np.logspace(-3, 4, 18)      # very different scales side by side
    O = np.column_stack([ox.xtv(R[:, v]) for v in range(18)])
    scale = np.sqrt(n) * np.abs(R).max(axis=0)
    out = {}
    try:
        for mode, tol in ((0, 2e-15), (4910, 2e-15), (1316, 2e-15), (428, 2e-15), (4908, 1e-12), (1308, 2e-7)):
            mih.set_xtv_digits(mode)
            got = x.xtv(R)
            assert np.all(np.max(np.abs(got - O), axis=0) < tol * scale + 1e-13 * np.abs(O).max(axis=0)), mode
            for m_rhs in (1, 2, 3, 4, 5, 7, 8, 10, 13, 15, 16, 17):   # 1 .. 9 operands: every pass split, 6-operand passes with and without a half-empty last operand
                assert np.array_equal(x.xtv(R[:, :m_rhs]), got[:, :m_rhs]), (mode, m_rhs)
            out[mode] = got
    finally:
        mih.set_xtv_digits(0)
    assert np.array_equal(out[0], out[4910])
    for mode in (1316, 428):
        assert np.all(np.max(np.abs(out[0] - out[mode]), axis=0) <= 2e-15 * scale), mode
    assert not np.array_equal(out[0], out[4908]) and not np.array_equal(out[0], out[1308])

def test_maf_weights(mih, normal_pair):
    """test/utilities_test.jl:215-232."""
    x, ox = normal_pair
    bed = mih.read_bed(os.path.join(FIX, "normal.bed"), x.n)
    code = np.unpackbits(bed[:2], axis=1, bitorder="little").reshape(2, -1, 2)[:, :x.n, :]
    code = code[:, :, 0] + 2 * code[:, :, 1]
    w = mih.maf_weights(x)
    assert np.all(w >= 1.0)
    for j in range(2):
        ok = code[j] != 1
        f = np.select([code[j] == 2, code[j] == 3], [1.0, 2.0], 0.0)[ok].sum() / (2 * ok.sum())
        m = min(f, 1 - f)
        assert w[j] == pytest.approx(1 / (2 * np.sqrt(m * (1 - m))), rel=1e-12)
    w2 = mih.maf_weights(x, max_weight=2.0)
    assert np.all((w2 >= 1.0) & (w2 <= 2.0))
    res = mih.fit_iht(np.loadtxt(os.path.join(FIX, "normal_y_fam6.txt")), x, None, k=5, weight=w2, verbose=False)
    assert np.count_nonzero(res.beta) == 5

def test_xtv_extreme_residual_scales(mih, oracle, normal_pair):
    """The fixed-point scale 2^e follows max|r|: huge, tiny, denormal and all-zero residuals stay finite and accurate."""
    x, ox = normal_pair
    r = np.random.default_rng(8).standard_normal(x.n)
    base = ox.xtv(r)
    for scale in (1e150, 1e-150, 1e-290, 5e-310):
        out = x.xtv(r * scale)
        assert np.all(np.isfinite(out))
        assert rel(out, base * scale) < (1e-10 if scale > 1e-300 else 1e-3), scale
    assert np.all(x.xtv(np.zeros(x.n)) == 0.0)

def test_group_norms_round_the_square_then_the_sum(mih, oracle):
    """project_group_sparse! ranks the groups by `group_norm[n] + y[j]^2` (utilities.jl:626): the square is rounded, then the sum.
    Fused into an fma -- what the HIP compiler did to the device kernel until round 4 (seed 9079 of tools/fuzz_parity.py found it)
    -- 2.2^2 + 1.8^2 + 1.3^2 comes out as 9.77 instead of 9.770000000000001 and ties with 2.0^2 + 1.7^2 + 1.2^2 + 1.2^2 = 9.77, and the
    tie goes to the group with the lower label.  Known answer: the group with the larger (unfused) norm survives J = 1."""
    y = np.array([-2.0, 1.7, -1.2, 1.2, 2.2, -1.8, 1.3, 0.05])
    group = np.array([1, 1, 1, 1, 2, 2, 2, 3])
    assert (2.2 * 2.2 + 1.8 * 1.8) + 1.3 * 1.3 > ((2.0 * 2.0 + 1.7 * 1.7) + 1.2 * 1.2) + 1.2 * 1.2       # 9.770000000000001 > 9.77
    want = np.array([0, 0, 0, 0, 2.2, -1.8, 1.3, 0])
    for k in (4, np.array([4, 3, 1])):
        assert np.array_equal(oracle.project_group_sparse(y, group, 1, k), want)
        assert np.array_equal(mih.project_group_sparse(y, group, 1, k), want)
    # embedded in a long vector (several blocks of the device sorts), the two groups scattered
    rng = np.random.default_rng(3)
    n = 5000
    big = np.round(rng.standard_normal(n) * 0.1, 2)
    grp = rng.integers(3, 40, n)
    pos = rng.choice(n, 7, replace=False)
    big[pos] = y[:7]; grp[pos] = group[:7]
    kk = np.full(39, 2); kk[0], kk[1] = 4, 3
    got, ref = mih.project_group_sparse(big, grp, 2, kk), oracle.project_group_sparse(big, grp, 2, kk)
    assert np.array_equal(got, ref)
    assert np.array_equal(got[pos[4:7]], y[4:7]) and np.count_nonzero(got[pos[:4]]) == 4      # both survive J = 2: ranks 1 and 2, in that order

@pytest.mark.parametrize("n", [40_000_000, 6_000_000])
def test_forty_million_rows(mih, oracle, n):
    """n = 40 000 000 samples (x 64 SNPs): more rows than 16 exact row slices of the FP6 (2^18 rows) and base-13
    (2^20) residual formats hold, so the default steps down to base-4 digits and raises the number of slices to
    keep the f32 accumulators exact; 32-bit row indices and 64-bit offsets at scale.  n = 6 000 000: the
    intermediate step (base-13 digits)."""
    p = 64
    x = mih.SnpLinAlg.synthetic(n, p, seed=99, missing_rate=0.001)
    rng = np.random.default_rng(9)
    r = rng.standard_normal(n)
    out = x.xtv(r)
    if n < 2 ** 24:
        mih.set_xtv_digits(1316)                                     # what the default stepped down to
        try:
            assert np.array_equal(x.xtv(r), out)
        finally:
            mih.set_xtv_digits(0)
        R3 = np.column_stack([r, -2.0 * r, r[::-1]])
        O3 = x.xtv(R3)
        assert np.array_equal(O3[:, 0], out) and np.array_equal(O3[:, 1], -2.0 * out)
    ox = oracle.Mat.from_bed_columns(x.export_bed()[:8], n)          # the first 8 columns on the CPU
    ref = ox.xtv(r)
    assert rel(out[:8], ref) < 1e-10
    assert np.array_equal(x.xtv(r), out)
    idx = np.array([1, 5]); val = np.array([0.7, -1.1])
    mask = np.zeros(8, np.uint8); mask[idx] = 1
    coef = np.zeros(8); coef[idx] = val
    assert rel(x.xv_sparse(idx, val), ox.xv_masked(mask, coef)) < 1e-11

@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_xtv_shapes(mih, dtype):
    """Dense X'r over row counts around the 256-load step of the LDS-staged kernel (one step, exact multiples, ragged
    tails, many steps), odd / non-multiple-of-4 row counts (the fallback kernels) and column counts that leave idle
    waves in the last block; both storage types, several right-hand sides, run-to-run reproducible."""
    rng = np.random.default_rng(31)
    for n in (2, 4, 510, 512, 516, 1024, 1028, 3000, 4100, 501, 1026):
        for p in (1, 3, 4, 9):
            X = rng.standard_normal((n, p)).astype(dtype)
            xd = mih.DenseMatrix(X)
            R = rng.standard_normal((n, 15))
            want = X.astype(np.float64).T @ R
            got = xd.xtv(R)                                             # fused passes of 8 (f64) / 4 + 4 + 2 + 1 residuals
            assert rel(got, want) < 1e-12, (n, p)
            assert np.array_equal(xd.xtv(R), got)
            assert np.array_equal(xd.xtv(R[:, 1]), got[:, 1])           # fused == single, bit for bit
            assert np.array_equal(xd.xtv(R[:, 2:5]), got[:, 2:5])
            assert np.array_equal(xd.xtv(R[:, 3:11]), got[:, 3:11])

def test_mirror_accepts_any_array_layout(mih):
    """The host mirror hands the library contiguous Float64 / Int64 / UInt8 buffers whatever it is given: C- or Fortran-ordered
    and strided covariates, strided / list / column-vector responses, genotype columns out of a strided view, paths as ranges
    or Int32 arrays, folds as lists or floats, weights, groups and train masks in other dtypes -- always the same model, bit
    for bit (a C-ordered z read as column-major would be a silently different design)."""
    rng = np.random.default_rng(0)
    n, p = 500, 150
    cols = make_bed(rng, n, p)
    x = mih.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)
    z = np.column_stack([np.ones(n), rng.standard_normal(n), rng.standard_normal(n)])
    y = rng.standard_normal(n) + 0.5 * z[:, 1]
    base = mih.fit_iht(y, x, np.asfortranarray(z), k=4, verbose=False)
    assert base.c[1] == pytest.approx(0.5, abs=0.15)                       # the covariate was read as the covariate
    same = lambda res, ref=base: np.array_equal(res.beta, ref.beta) and np.array_equal(res.c, ref.c)
    big = np.zeros((n, 6)); big[:, ::2] = z
    yy = np.zeros(2 * n); yy[::2] = y
    cols_big = np.zeros((p, cols.shape[1] * 2), dtype=np.uint8); cols_big[:, ::2] = cols
    x2 = mih.SnpLinAlg(cols_big[:, ::2], n=n, center=True, scale=True, impute=True)
    for name, res in (("z C-order", mih.fit_iht(y, x, np.ascontiguousarray(z), k=4, verbose=False)),
                      ("z strided", mih.fit_iht(y, x, big[:, ::2], k=4, verbose=False)),
                      ("z nested lists", mih.fit_iht(y, x, z.tolist(), k=4, verbose=False)),
                      ("y strided", mih.fit_iht(yy[::2], x, z, k=4, verbose=False)),
                      ("y list", mih.fit_iht(y.tolist(), x, z, k=4, verbose=False)),
                      ("y column vector", mih.fit_iht(y.reshape(-1, 1), x, z, k=4, verbose=False)),
                      ("strided genotype columns", mih.fit_iht(y, x2, z, k=4, verbose=False))):
        assert same(res), name
    folds = hash_folds(n, 3)
    a = mih.cv_iht(y, x, z, path=[1, 2, 3], q=3, folds=folds, verbose=False)
    assert np.array_equal(a, mih.cv_iht(y, x, z, path=range(1, 4), q=3, folds=folds.astype(np.int64).tolist(), verbose=False))
    assert np.array_equal(a, mih.cv_iht(y, x, z, path=np.array([1, 2, 3], dtype=np.int32), q=3, folds=folds.astype(np.float64), verbose=False))
    Y = np.vstack([y, rng.standard_normal(n)])
    m1 = mih.fit_iht(Y, x, z.T.copy(), k=4, verbose=False)
    assert np.array_equal(m1.beta, mih.fit_iht(np.asfortranarray(Y), x, np.asfortranarray(z.T), k=4, verbose=False).beta)
    assert np.array_equal(m1.beta, mih.fit_iht(np.ascontiguousarray(Y), x, np.ascontiguousarray(z.T), k=4, verbose=False).beta)
    w = rng.uniform(0.5, 2, p); wbig = np.zeros(2 * p); wbig[::2] = w
    w1 = mih.fit_iht(y, x, z, k=4, weight=w, verbose=False)
    assert same(mih.fit_iht(y, x, z, k=4, weight=wbig[::2], verbose=False), w1) and same(mih.fit_iht(y, x, z, k=4, weight=w.tolist(), verbose=False), w1)
    g = (np.arange(p) % 5 + 1)
    g1 = mih.fit_iht(y, x, z, k=2, J=2, group=g, verbose=False)
    assert same(mih.fit_iht(y, x, z, k=2, J=2, group=g.astype(np.int32), verbose=False), g1) and same(mih.fit_iht(y, x, z, k=2, J=2, group=g.tolist(), verbose=False), g1)
    t = rng.random(n) < 0.8
    t1 = mih.fit_iht(y, x, z, k=4, train=t, verbose=False)
    assert same(mih.fit_iht(y, x, z, k=4, train=t.astype(np.uint8), verbose=False), t1) and same(mih.fit_iht(y, x, z, k=4, train=t.astype(np.int64), verbose=False), t1)

def test_xtv_accuracy_against_exact_rational_arithmetic(mih):
    """The fixed-point X'r against EXACT dot products (Python rationals) of the raw dosages: the only rounding is that
    of the residual to 2^-55 max|r| (2^-58 in the base-13 format) plus the recombination in f64, so the error stays
    at a few 1e-16 of sqrt(n) max|r| -- well inside what an n-term f64 dot product guarantees (n 2^-53 sum|g r|)."""
    from fractions import Fraction

    rng = np.random.default_rng(2718)
    n, p = 3000, 40
    cols = make_bed(rng, n, p, maf_lo=0.05)
    x = mih.SnpLinAlg(cols, n=n, center=False, scale=False, impute=False)
    bits = np.unpackbits(cols, axis=1, bitorder="little").reshape(p, -1, 2)[:, :n, :]
    code = bits[:, :, 0] + 2 * bits[:, :, 1]
    g = np.select([code == 0, code == 2, code == 3], [0, 1, 2], default=0)            # missing (code 1) counts as 0
    r = rng.standard_normal(n) * np.exp(rng.uniform(-6, 6, n))                       # 5 decades of dynamic range
    rf = [Fraction(float(v)) for v in r]
    exact = [sum((int(gi) * ri for gi, ri in zip(g[j], rf) if gi), Fraction(0)) for j in range(p)]
    scale = np.sqrt(n) * np.abs(r).max()
    f64_bound = n * 2.0 ** -53 * (g * np.abs(r)).sum(axis=1)
    try:
        for mode, tol in ((0, 6e-16), (1316, 2e-16), (428, 6e-16)):
            mih.set_xtv_digits(mode)
            got = x.xtv(r)
            err = np.array([abs(float(Fraction(float(got[j])) - exact[j])) for j in range(p)])
            assert err.max() <= tol * scale, (mode, err.max() / scale)
            assert np.all(err <= f64_bound), mode
    finally:
        mih.set_xtv_digits(0)

@pytest.mark.parametrize("shape", ["one_outlier_1e8", "one_outlier_1e12", "two_outliers", "twelve_decades", "cauchy"])
def test_xtv_fixed_point_under_adversarial_dynamic_range(mih, shape):
    """ X'r stays f64-grade whatever the residual looks like.  The fixed point keeps 54 bits of the LARGEST
    entry it carries; rows that tower over the rest (max|r| > 64 x the lower quartile of the 256-row block maxima, at most 64 of
    them: csrc/peel.h) leave it and ride an f64 side channel in k_xtv_finalize, so the scale is set by the rest.  Against EXACT
    rational dot products, for the three residual formats:
      * one entry 1e8 / 1e12 x the rest, two outliers of different size: ONE residual peeled (counter), every column -- with
        or without the outlier -- within 2 ulp-sums (2 x 2^-53 sum_i g_ij |r_i|; numpy's pairwise sum is held to 8) and within
        1e-13 of its own value where that value has not cancelled (round 5: 2e-7 on the columns without the outlier);
      * twelve decades, log-uniform: no outlier by the guard's rule, nothing peeled, 8 ulp-sums like numpy's (as in round 5);
      * a Cauchy residual (a heavy tail rather than a few outliers): whether or not the guard peels the extreme row, the result
        is within 32 ulp-sums."""
    rng = np.random.default_rng(31415)
    n, p = 3000, 48
    cols = make_bed(rng, n, p, maf_lo=0.05)
    x = mih.SnpLinAlg(cols, n=n, center=False, scale=False, impute=False)
    g = _dosages(cols, n)
    r = rng.standard_normal(n)
    i0 = int(np.argmax(np.abs(r)))
    if shape == "one_outlier_1e8":
        r[i0] *= 1e8
    elif shape == "one_outlier_1e12":
        r[i0] *= 1e12
    elif shape == "two_outliers":
        r[5] *= 1e9
        r[2000] *= -3e6
    elif shape == "twelve_decades":
        r = rng.standard_normal(n) * 10.0 ** rng.uniform(-12, 0, n)
    else:
        r = rng.standard_cauchy(n)
    exact = _exact_xtv(g, r)
    from fractions import Fraction
    ex = np.array([float(e) for e in exact])
    pairwise = np.array([np.sum(g[j].astype(np.float64) * r) for j in range(p)])
    err_np = np.array([abs(float(Fraction(float(pairwise[j])) - exact[j])) for j in range(p)])
    ulp_sums = 2.0 ** -53 * (g * np.abs(r)).sum(axis=1)
    assert np.all(err_np <= 8 * ulp_sums)
    outliers = shape in ("one_outlier_1e8", "one_outlier_1e12", "two_outliers")
    mih.profile_enable(x, True)
    try:
        for mode in (0, 428, 1316):
            mih.set_xtv_digits(mode)
            mih.profile_counters(x, reset=True)
            got = x.xtv(r)
            peeled = mih.profile_counters(x, reset=True)["peeled_residuals"]
            err = np.array([abs(float(Fraction(float(got[j])) - exact[j])) for j in range(p)])
            if outliers:
                assert peeled == 1, (mode, peeled)
                assert np.all(err <= 2 * ulp_sums), (mode, float((err / ulp_sums).max()))
                rel = err / np.abs(ex)
                assert rel.max() <= 1e-12 and rel[np.abs(ex) >= 1.0].max() <= 1e-13, (mode, rel.max())
            elif shape == "twelve_decades":
                assert peeled == 0, (mode, peeled)
                assert np.all(err <= 8 * ulp_sums), (mode, float((err / ulp_sums).max()))
            else:
                assert np.all(err <= 32 * ulp_sums), (mode, float((err / ulp_sums).max()))
            # the guard's decision against its restatement in numpy (tests/gpu_helpers.py): the same rows, whatever the kernel shape
            assert peeled == (1 if peel_rule(r).size else 0), (shape, mode, peeled, peel_rule(r))
    finally:
        mih.set_xtv_digits(0)
        mih.profile_enable(x, False)

def test_peeled_rows_in_fused_passes_with_missing_genotypes(mih, oracle):
    """The side channel inside fused multi-residual passes and on a matrix with imputed entries: 23 residuals in one call (two
    passes of the flat packing), some with planted outliers -- one of them on a row where genotypes are missing --, some without.
    (1) every residual against the oracle's f64 dot products; (2) a residual WITHOUT an outlier gives the bits it gives alone and
    in any company (the guard looks at its own block maxima only); (3) the counter says which residuals were peeled; (4) more
    than 64 rows above the guard's threshold: no peel, the plain scale (the result of round 5, to its documented accuracy)."""
    n, p = 6001, 700
    x = mih.SnpLinAlg.synthetic(n, p, seed=5, missing_rate=0.02)
    cols = x.export_bed()
    ox = oracle.Mat.from_bed_columns(cols, n)
    codes = np.unpackbits(cols, axis=1, bitorder="little").reshape(p, -1, 2)[:, :n, :]
    miss_rows = np.flatnonzero(((codes[:, :, 0] == 1) & (codes[:, :, 1] == 0)).any(axis=0))
    assert miss_rows.size > 100
    rng = np.random.default_rng(99)
    m = 23
    R = rng.standard_normal((m, n))
    planted = {2: [(int(miss_rows[7]), 3e9)], 5: [(17, -1e7), (4000, 2e11)], 11: [(int(i), 1e6 * (1 + t)) for t, i in enumerate(rng.choice(n, 40, replace=False))],
               20: [(n - 1, 5e8)]}
    for v, lst in planted.items():
        for i, f in lst:
            R[v, i] *= f
    heavy = 14                                    # 200 rows 1e6 x the rest: beyond the side channel's 64
    R[heavy, rng.choice(n, 200, replace=False)] *= 1e6
    mih.profile_enable(x, True)
    mih.profile_counters(x, reset=True)
    got = x.xtv(R.T).T
    assert mih.profile_counters(x, reset=True)["peeled_residuals"] == len(planted)
    for v in range(m):                               # ... and exactly the planted rows, by the rule's restatement (tests/gpu_helpers.py)
        want_rows = sorted(i for i, _ in planted.get(v, []))
        assert sorted(peel_rule(R[v]).tolist()) == want_rows, (v, peel_rule(R[v]), want_rows)
    for v in range(m):
        want = ox.xtv(R[v])
        scale = np.abs(want) + 1e-3 * np.abs(want).max()
        tol = 1e-6 if v == heavy else 1e-11          # (the oracle is a plain f64 loop over 6001 terms: ~1e-12 of its own)
        assert np.all(np.abs(got[v] - want) <= tol * scale), (v, float((np.abs(got[v] - want) / scale).max()))
    plain = [v for v in range(m) if v not in planted and v != heavy]
    alone = x.xtv(R[plain[:3]].T).T
    for t, v in enumerate(plain[:3]):
        assert np.array_equal(alone[t].view(np.uint64), got[v].view(np.uint64)), v
    one = x.xtv(R[[5]].T).T
    assert np.array_equal(one[0].view(np.uint64), got[5].view(np.uint64))          # ... and a peeled one too
    mih.profile_enable(x, False)

def test_naive_impute(mih, tmp_path):
    """naive_impute (src/utilities.jl:862-899): missing entries -> the SNP's most frequent genotype, ties resolved in the
    order of the reference's if / elseif chain (0x02, then 0x03, then 0x00); every other entry and the file header are
    unchanged.  Bit-exact against a direct numpy statement of that loop."""
    rng = np.random.default_rng(862)
    n, p = 1003, 257
    cols = make_bed(rng, n, p, missing_rate=0.07)
    code = np.stack([(cols[:, i // 4] >> (2 * (i % 4))) & 3 for i in range(n)], axis=1)          # p x n PLINK codes
    # force ties: column 5 gets equal 0x00 and 0x02 counts, column 6 equal 0x02 and 0x03, column 7 equal 0x00 and 0x03
    for j, (a, b) in ((5, (0, 2)), (6, (2, 3)), (7, (0, 3))):
        code[j, :] = 1
        code[j, 0:300] = a
        code[j, 300:600] = b
        code[j, 600:650] = ({0, 2, 3} - {a, b}).pop()
    padded = np.zeros((p, ((n + 3) // 4) * 4), dtype=np.uint8)
    padded[:, :n] = code
    cols = (padded[:, 0::4] | (padded[:, 1::4] << 2) | (padded[:, 2::4] << 4) | (padded[:, 3::4] << 6)).astype(np.uint8)
    want = code.copy()
    for j in range(p):
        e0, e1, e2 = (code[j] == 0).sum(), (code[j] == 2).sum(), (code[j] == 3).sum()
        most = max(e0, e1, e2)
        fill = 2 if most == e1 else 3 if most == e2 else 0
        want[j, code[j] == 1] = fill
    assert want[5, 700] == 2 and want[6, 700] == 2 and want[7, 700] == 3                       # the tie rules fired
    dest = tmp_path / "imputed.bed"
    mih.naive_impute(cols, str(dest), n=n)
    raw = np.fromfile(dest, dtype=np.uint8)
    assert bytes(raw[:3]) == b"\x6c\x1b\x01" and raw.size == 3 + p * ((n + 3) // 4)
    got = raw[3:].reshape(p, -1)
    gcode = np.stack([(got[:, i // 4] >> (2 * (i % 4))) & 3 for i in range(n)], axis=1)
    assert np.array_equal(gcode, want)
    assert not np.any(gcode == 1)
    if n % 4:                                                                                   # padding bits of the last byte stay 0
        assert np.all(got[:, -1] >> (2 * (n % 4)) == 0)
    # a SnpLinAlg built from the imputed file has no missing entries and the same non-missing genotypes
    x2 = mih.SnpLinAlg(mih.read_bed(str(dest), n), n)
    assert np.array_equal(x2.export_bed(), got)

def test_concurrent_fits_with_different_digit_formats(mih):
    """The library has no process-wide kernel or format selector: the residual format travels with the call.
    Two host threads fitting CONCURRENTLY on one shared matrix, one in the default 54-bit format and one in the 43-bit fast
    format (plus a cross-validation in a third), must give the bits of the same calls run one after the other."""
    import threading
    n, p = 12_000, 3_000
    x = mih.SnpLinAlg.synthetic(n, p, seed=77, missing_rate=0.01)
    rng = np.random.default_rng(78)
    supp = np.sort(rng.choice(p, 12, replace=False))
    eta = x.xv_sparse(supp, rng.standard_normal(12) * 0.5)
    Y = np.vstack([eta + rng.standard_normal(n), 0.5 * eta + rng.standard_normal(n), rng.standard_normal(n)])
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    folds = hash_folds(n, 3)
    jobs = {
        "mv_default": lambda: mih.fit_iht(Y, x, None, k=20, verbose=False, max_iter=30, xtv_digits=0).beta,
        "mv_fast": lambda: mih.fit_iht(Y, x, None, k=20, verbose=False, max_iter=30, xtv_digits=4908).beta,
        "cv_1316": lambda: mih.cv_iht(yb, x, None, path=range(1, 9), q=3, folds=folds, verbose=False, return_raw=True,
                                      d=mih.Bernoulli(), l=mih.LogitLink(), xtv_digits=1316)[1],
        "xtv_fast": lambda: x.xtv(Y.T.copy(), xtv_digits=4908),
        "xtv_default": lambda: x.xtv(Y.T.copy()),
    }
    serial = {k: f() for k, f in jobs.items()}
    assert not np.array_equal(serial["xtv_fast"], serial["xtv_default"])        # the formats really differ
    for _ in range(3):
        out, errs = {}, []

        def run(name):
            try:
                out[name] = jobs[name]()
            except Exception as e:                                              # noqa: BLE001
                errs.append((name, e))
        th = [threading.Thread(target=run, args=(k,)) for k in jobs]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for k in jobs:
            assert np.array_equal(np.asarray(out[k]).view(np.uint64), np.asarray(serial[k]).view(np.uint64)), k

def test_ingest_pipeline_at_full_row_count(mih):
    """mih_snp_create's upload pipeline (round 3: eight workers with their own streams pulling 16 MB chunks of whole column groups
    from one queue) at the row count of the benchmark: 3000 columns of n = 500 000 (375 MB, 24 chunks, a ragged last one) with
    missing genotypes must give the matrix the on-device generator built -- same bytes back out, same column statistics, same
    X'r bits."""
    n, p = 500_000, 3000
    xs = mih.SnpLinAlg.synthetic(n, p, seed=31, missing_rate=0.01)
    cols = xs.export_bed()
    x = mih.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)
    assert np.array_equal(x.export_bed(), cols)
    mu_s, sv_s = xs.mu_sigma()
    mu, sv = x.mu_sigma()
    assert np.array_equal(mu, mu_s) and np.array_equal(sv, sv_s)
    r = np.random.default_rng(3).standard_normal(n)
    assert np.array_equal(x.xtv(r), xs.xtv(r))
    # a strided source (col_stride_bytes > ceil(n/4)) through the same pipeline
    wide = np.zeros((p, cols.shape[1] + 37), dtype=np.uint8)
    wide[:, :cols.shape[1]] = cols
    x2 = mih.SnpLinAlg(wide, n=n, center=True, scale=True, impute=True)
    assert np.array_equal(x2.export_bed(), cols)

def test_ingest_of_a_tall_matrix_stays_within_the_staging_budget(mih):
    """ A chunk of the upload pipeline is at least one group of 32 columns, so above a 512 KB column stride it outgrows
    the 16 MB target: at n = 4.4M rows (1.1 MB per column) a chunk is 35 MB and eight workers with two buffers each would pin 560 MB
    of host memory and take as much VRAM; mih_snp_create caps the staging of all workers at 512 MB (fewer workers) and degrades
    to one worker if the allocation fails.  The matrix must be the one the on-device generator builds: same bytes back out, same
    column statistics, same X'r bits (n > 2^22 rows: the fused formats step down, §3.1), incl. a strided source and a ragged
    last chunk (70 columns = 32 + 32 + 6)."""
    n, p = 4_400_000, 70
    xs = mih.SnpLinAlg.synthetic(n, p, seed=77, missing_rate=0.002)
    cols = xs.export_bed()
    assert cols.shape == (p, (n + 3) // 4)
    x = mih.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)
    assert np.array_equal(x.export_bed(), cols)
    mu_s, sv_s = xs.mu_sigma()
    mu, sv = x.mu_sigma()
    assert np.array_equal(mu, mu_s) and np.array_equal(sv, sv_s)
    r = np.random.default_rng(4).standard_normal(n)
    got = x.xtv(r)
    assert np.array_equal(got, xs.xtv(r))
    # column 3 against numpy (dosage codes 00 -> 0, 10 -> 1, 11 -> 2, 01 -> missing = the column mean)
    j = 3
    code = np.stack([(cols[j] >> (2 * t)) & 3 for t in range(4)], axis=1).ravel()[:n]
    g = np.array([0.0, np.nan, 1.0, 2.0])[code]
    g[np.isnan(g)] = mu[j]
    assert got[j] == pytest.approx(float(np.dot((g - mu[j]) * sv[j], r)), rel=1e-10)
    wide = np.zeros((p, cols.shape[1] + 5), dtype=np.uint8)
    wide[:, :cols.shape[1]] = cols
    assert np.array_equal(mih.SnpLinAlg(wide, n=n, center=True, scale=True, impute=True).export_bed(), cols)


@pytest.mark.parametrize("center", [True, False])
def test_xtv_with_a_non_finite_residual(mih, oracle, center):
    """A NaN or an Inf in the residual: the reference's `mul!` is a floating-point sum (utilities.jl:133) and hands the value on -- NaN or
    +-Inf in every column that touches the entry.  The fixed point has no such value: the affected residual's row of the result is NaN
    in every column (include/mendeliht_hip.h), centered matrix or not, whatever the format; the other residuals of the same fused
    pass are what they are without it."""
    rng = np.random.default_rng(99)
    n, p = 1300, 70
    cols = make_bed(rng, n, p, missing_rate=0.01)
    x = mih.SnpLinAlg(cols, n=n, center=center, scale=center, impute=True)
    ox = oracle.Mat.from_bed_columns(cols, n, center=center, scale=center, impute=True)
    R = rng.standard_normal((5, n))
    clean = x.xtv(R.T).T
    bad = R.copy()
    bad[1, 17] = np.nan
    bad[3, 1200] = np.inf
    try:
        for mode in (0, 428, 1316):
            mih.set_xtv_digits(mode)
            got = x.xtv(bad.T).T
            assert np.all(np.isnan(got[1])) and np.all(np.isnan(got[3])), mode
            for v in (0, 2, 4):
                assert np.array_equal(got[v], x.xtv(R[v])), (mode, v)
            assert np.all(np.isnan(x.xtv(bad[1]))) and np.all(np.isnan(x.xtv(-bad[3]))), mode      # the single-residual kernel
    finally:
        mih.set_xtv_digits(0)
    for v in (0, 2, 4):
        assert rel(clean[v], ox.xtv(R[v])) < 1e-12
    with np.errstate(invalid="ignore"):
        assert not np.any(np.isfinite(ox.xtv(bad[1])[np.abs(ox.xtv(np.eye(n)[17])) > 0]))      # the oracle's floating-point sums hand the NaN on
