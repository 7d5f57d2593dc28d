"""Helpers shared by the GPU test files (split out of test_gpu_parity.py in round 6)."""
import os

import numpy as np

from conftest import hash_folds


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / (np.max(np.abs(b)) + 1e-300))

def _run_probe_snippet(snippet, out_file, extra_env=None, probes=True, timeout=900):
    """A python snippet in its own process, on the measurement build of the library (MENDELIHT_HIP_PROBES=1) or on the product."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "MENDELIHT_HIP_PROBES"}
    if probes:
        env["MENDELIHT_HIP_PROBES"] = "1"
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", snippet, root, str(out_file)], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(out_file)

def _sim(oracle, ox, rng, k, scale=0.5):
    p = ox.p
    b = np.zeros(p)
    supp = rng.choice(p, k, replace=False)
    b[supp] = rng.standard_normal(k) * scale
    mask = np.zeros(p, np.uint8)
    mask[supp] = 1
    return ox.xv_masked(mask, b)

def _mv_problem(oracle, ox, rng, r, k, q=1):
    p, n = ox.p, ox.n
    B = np.zeros((r, p))
    for _ in range(k):
        B[rng.integers(r), rng.integers(p)] = rng.standard_normal() * 0.6
    XB = np.zeros((r, n))
    for i in range(r):
        mask = (B[i] != 0).astype(np.uint8)
        XB[i] = ox.xv_masked(mask, B[i])
    A = rng.standard_normal((r, r))
    L = np.linalg.cholesky(A @ A.T / r + np.eye(r) * 0.5)
    Z = np.vstack([np.ones(n)] + [rng.standard_normal(n) for _ in range(q - 1)])
    Cm = rng.standard_normal((r, q))
    Y = XB + Cm @ Z + L @ rng.standard_normal((r, n))
    return Y, Z

_BT_TIE = 1e-13      # orc_result.bt_cond below this: `old_logl > new_logl` compared two sums of n terms that agree to their rounding error

_NUDGES = [1.0 + e * 2.0 ** -51 for e in (2, 1, 3, 4, 6, 8)]      # a few ulps: one nudge can land on the same branch by luck (seed 2449)

def _unstable(a, b, rtol, atol=1e-10):
    """The oracle against ITSELF on covariates scaled by 1 + a few 2^-51 (a: the run on the original input, b: a nudged one, dicts
    of arrays / scalars): True when an ulp-sized change of the input moves the oracle's own answer by more than the tolerance.
    Such a trajectory amplifies rounding from step to step (seed 2121 of tools/fuzz_parity.py: the intercepts of a multivariate
    fit drift apart by x1.87 per iteration, 1e-15 -> 4e-7 over 40 steps; seed 2275: a Bernoulli fit that backtracks three
    times in most steps) -- no two floating-point implementations agree on it, the reference under another BLAS included,
    so the sweeps do not hold the GPU to it."""
    for key in a:
        va, vb = np.asarray(a[key], dtype=float), np.asarray(b[key], dtype=float)
        if va.shape != vb.shape or not np.allclose(va, vb, rtol=rtol, atol=atol):
            return True
    return False

def _rows_permuted(oracle, ox, perm):
    """The oracle's matrix with its rows in another order.  With y, z and every per-sample option permuted alike the PROBLEM is the same
    one -- same columns, same means and scales (integer counts), same optimum -- but every sum over the samples (X'r, the loglikelihood,
    the step size, debias!'s normal equations) adds its terms in another order: exactly the freedom another correct implementation
    of the reference has (a threaded BLAS, this library's tiles).  A trajectory the oracle does not reproduce under that is one
    nobody can be held to (the set-aside classes that were accepted by argument)."""
    keep, n = ox._keep, ox.n
    if keep.dtype != np.uint8:
        return oracle.Mat.from_dense(np.asarray(keep)[perm])
    code = np.stack([(keep >> s) & 3 for s in (0, 2, 4, 6)], axis=2).reshape(keep.shape[0], -1)[:, :n][:, perm]
    pad = np.zeros((keep.shape[0], keep.shape[1] * 4), np.uint8)
    pad[:, :n] = code
    return oracle.Mat.from_bed_columns(pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6), n)

def _row_orders(n, count=4):
    return [np.random.default_rng(7700 + t).permutation(n) for t in range(count)]

def _within_own_spread(o, own, got, tols, factor=4.0):
    """o: the oracle's answer, own: its answers to the same problem re-associated (ulp nudges of z, other row orders), got: the
    device's; dicts of arrays / scalars over the keys of tols = {key: (rtol, atol)}.  True when the device is within the tolerance
    PLUS factor x the oracle's own spread, entry by entry, and that spread is not nil: a trajectory the oracle reproduces to within
    the tolerance but not by much (seed 16276 of tools/fuzz_parity.py: an effect of 0.008 moves by 8e-7 from one row order to the
    next, the device's is 1.5e-6 away, the tolerance is 8e-7)."""
    any_spread = False
    for key, (rtol, atol) in tols.items():
        ref = np.asarray(o[key], dtype=float)
        spread = np.max([np.abs(np.asarray(v[key], dtype=float) - ref) for v in own], axis=0)
        any_spread |= bool(np.max(spread) > 0.0)
        if np.shape(got[key]) != ref.shape or not np.all(np.abs(np.asarray(got[key], dtype=float) - ref) <= rtol * np.abs(ref) + atol + factor * spread):
            return False
    return any_spread

def _exact_xtv(g, r):
    from fractions import Fraction
    rf = [Fraction(float(v)) for v in r]
    return [sum((int(gi) * ri for gi, ri in zip(g[j], rf) if gi), Fraction(0)) for j in range(g.shape[0])]

def _dosages(cols, n):
    p = cols.shape[0]
    bits = np.unpackbits(cols, axis=1, bitorder="little").reshape(p, -1, 2)[:, :n, :]
    code = bits[:, :, 0] + 2 * bits[:, :, 1]
    return np.select([code == 0, code == 2, code == 3], [0, 1, 2], default=0)

def _same_fit(a, b, what):
    assert a.iter == b.iter, (what, a.iter, b.iter)
    assert list(a.trace["backtracks"]) == list(b.trace["backtracks"]), what
    assert np.array_equal(np.flatnonzero(a.beta), np.flatnonzero(b.beta)), what
    # every sum of the resident chain is formed in the host-driven kernels' order; only the scalar log / lgamma of the
    # loglikelihood's closed form comes from another libm (device against host): the last bit of the trace may differ
    np.testing.assert_allclose(a.trace["logl"], b.trace["logl"], rtol=4e-16, atol=0, err_msg=what)
    assert np.array_equal(a.trace["tol"], b.trace["tol"]), what
    assert np.array_equal(a.beta, b.beta) and np.array_equal(a.c, b.c), what
    assert np.array_equal(a.mu, b.mu), what
    assert a.choose_fired == b.choose_fired, what

def _config3_problem(mih, n, p, seed=2024):
    """BASELINE configs[3] in small: Bernoulli/Logit response with 10 true effects on a synthetic SnpArray, explicit hash folds."""
    x = mih.SnpLinAlg.synthetic(n, p, seed=seed)
    rng = np.random.default_rng(2025)
    supp = np.sort(rng.choice(p, 10, replace=False))
    eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    return x, yb, hash_folds(n, 5)

def _config4_problem(x, rng, r, k, lo=0.15, hi=0.45):
    """r traits on matrix x with k planted effects spread over the traits (each trait its own columns), an intercept per trait and
    errors with an AR(1) covariance: returns Y (r x n), the planted B (r x p) as {trait: (columns, effects)} and Sigma."""
    n, p = x.n, x.p
    lin = rng.choice(r * p, k, replace=False)
    Sigma = 0.5 ** np.abs(np.subtract.outer(np.arange(r), np.arange(r)))        # AR(1), rho = 0.5
    L = np.linalg.cholesky(Sigma)
    Y = L @ rng.standard_normal((r, n))
    planted = {}
    for t in range(r):
        cols = np.unique(lin[lin % r == t] // r)
        eff = rng.choice([-1.0, 1.0], cols.size) * rng.uniform(lo, hi, cols.size)
        planted[t] = (cols, eff)
        Y[t] += x.xv_sparse(cols, eff) + 1.0 + 0.1 * t
    return Y, planted, Sigma


def peel_rule(r, ratio=64.0, max_rows=64):
    """The outlier guard of the fixed-point residual restated (csrc/peel.h): max|r| per strided 256-row block (block b: rows
    256 b + t + 16384 j), bq = the ceil(B/4)-th smallest of the B non-empty blocks' maxima; the guard fires iff max|r| > ratio * bq;
    then the rows with |r_i| > ratio * bq are peeled if there are at most max_rows of them.  Returns the peeled row indices."""
    r = np.asarray(r, dtype=np.float64)
    n = r.size
    nb = min(64, (n + 255) // 256)
    a = np.abs(r)
    bmax = np.zeros(nb)
    for b in range(nb):
        rows = np.concatenate([np.arange(lo, min(lo + 256, n)) for lo in range(256 * b, n, 16384)])
        bmax[b] = a[rows].max()
    bq = np.sort(bmax)[(nb + 3) // 4 - 1]
    tau = ratio * bq
    if not (bmax.max() > tau) or not np.isfinite(bmax.max()):
        return np.zeros(0, dtype=np.int64)
    rows = np.flatnonzero(a > tau)
    return rows if rows.size <= max_rows else np.zeros(0, dtype=np.int64)

