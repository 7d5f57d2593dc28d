"""Replay single trials of the seeded sweeps (tests/test_gpu_parity.py) that a tools/fuzz_parity.py campaign flagged.
usage: python tools/repro_fuzz.py options SEED TRIAL | projections SEED"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mendeliht_amd as mih
from oracle import oracle
import test_gpu_parity as T

what, seed = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(seed)
if what == "options":
    want = int(sys.argv[3])
    for trial in range(want + 1):
        x, ox, y, z, k, kw, okw, both, tol, fam, tag = T._options_case(mih, oracle, rng, trial)
    print("tag", tag)
    o = oracle.fit_iht(ox, y, z, k=k, max_iter=40, **okw, **both)
    res = mih.fit_iht(y, x, z, k=k, max_iter=40, verbose=False, **kw, **both)
    print("oracle: iter", o["iter"], "logl", o["logl"], "r", o["nb_r"], "bt", o["bt_trace"].tolist(), "eta_cond", o["eta_cond"])
    print("gpu   : iter", res.iter, "logl", res.logl, "r", getattr(res.d, "r", None), "bt", res.trace["backtracks"].tolist())
    print("oracle logl trace", o["logl_trace"].tolist())
    print("gpu    logl trace", res.trace["logl"].tolist())
    for g in T._NUDGES:
        o2 = oracle.fit_iht(ox, y, z * g, k=k, max_iter=40, **okw, **both)
        print("nudge", g, "iter", o2["iter"], "logl", o2["logl"], "r", o2["nb_r"])
if what == "projections":
    n = int(rng.choice([1, 2, 3, 31, 64, 1000, 2047, 2049, 65535, 65537])) if rng.random() < 0.4 else int(rng.integers(1, 300000))
    v = rng.standard_normal(n) * 10.0 ** float(rng.integers(-3, 4))
    style = int(rng.integers(0, 6))
    if style == 1: v = np.round(v, int(rng.integers(0, 3)))
    elif style == 2: v = rng.choice([-2.5, -1.0, 0.0, 1.0, 2.5, 7.0], n)
    elif style == 3:
        v[rng.random(n) < 0.01] = np.inf; v[rng.random(n) < 0.01] = -np.inf
    elif style == 4: v *= 1e-310
    elif style == 5: v[rng.random(n) < 0.7] = 0.0
    k = int(rng.choice([1, min(2, n), n, max(1, n - 1), max(1, n // 2)])) if rng.random() < 0.4 else int(rng.integers(1, n + 1))
    G = int(rng.integers(1, min(n, 3000) + 1))
    group = np.sort(rng.integers(1, G + 1, n)) if rng.random() < 0.5 else rng.integers(1, G + 1, n)
    group[rng.integers(0, n)] = G
    J = int(rng.integers(1, G + 1))
    kg = rng.integers(0, 5, G) if rng.random() < 0.5 else int(rng.integers(1, 5))
    w = v.copy(); w[~np.isfinite(w)] = 1e6
    got, want = mih.project_group_sparse(w, group, J, kg), oracle.project_group_sparse(w, group, J, kg)
    bad = np.flatnonzero(got != want)
    print("n", n, "style", style, "G", G, "J", J, "vector k", np.ndim(kg) > 0, "differ at", bad.size, "positions")
    kk = (lambda g: kg) if np.ndim(kg) == 0 else (lambda g: kg[g - 1])
    # group norms as the reference computes them
    perm = np.argsort(-np.abs(w), kind="stable")
    cnt = np.zeros(G + 1, int); norm = np.zeros(G + 1)
    for j in perm:
        g = group[j]
        if cnt[g] < kk(g): norm[g] += w[j] ** 2; cnt[g] += 1
    order = np.argsort(-norm[1:], kind="stable")
    rank = np.empty(G, int); rank[order] = np.arange(1, G + 1)
    for j in bad[:12]:
        g = group[j]
        print(f"  index {j}: value {w[j]!r} group {g} k_g {kk(g)} norm {norm[g]!r} rank {rank[g - 1]} (J = {J}) gpu {got[j]!r} oracle {want[j]!r}; "
              f"groups with the same norm: {np.flatnonzero(norm[1:] == norm[g]) + 1}")
    for r in range(3):
        again = mih.project_group_sparse(w, group, J, kg)
        print("  rerun", r, "differs from the oracle at", np.flatnonzero(again != want).size, "positions; equal to the first GPU result:", np.array_equal(again, got, equal_nan=True))
