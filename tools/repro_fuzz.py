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
    def side(f):                                   # (either side may end in the reference's error: that is a result too)
        try:
            return f(), None
        except (RuntimeError, mih.MendelIHTError) as e:
            return None, str(e)
    o, oe = side(lambda: oracle.fit_iht(ox, y, z, k=k, max_iter=40, **okw, **both))
    res, re_ = side(lambda: mih.fit_iht(y, x, z, k=k, max_iter=40, verbose=False, **kw, **both))
    if o is None: print("oracle: ERROR", oe)
    else:
        print("oracle: iter", o["iter"], "logl", o["logl"], "r", o["nb_r"], "bt", o["bt_trace"].tolist(), "eta_cond", o["eta_cond"], "bt_cond", o["bt_cond"])
        print("oracle logl trace", o["logl_trace"].tolist())
        print("oracle support", np.flatnonzero(o["beta"]).tolist(), "c", o["c"].tolist())
    if res is None: print("gpu   : ERROR", re_)
    else:
        print("gpu   : iter", res.iter, "logl", res.logl, "r", getattr(res.d, "r", None), "bt", res.trace["backtracks"].tolist())
        print("gpu    logl trace", res.trace["logl"].tolist())
        print("gpu    support", np.flatnonzero(res.beta).tolist(), "c", res.c.tolist())
    if o is not None and res is not None:
        sb = np.flatnonzero((o["beta"] != 0) | (res.beta != 0))
        print("oracle beta", o["beta"][sb].tolist())
        print("gpu    beta", res.beta[sb].tolist())
        print("relative differences: beta", (np.abs(res.beta[sb] - o["beta"][sb]) / np.maximum(np.abs(o["beta"][sb]), 1e-300)).tolist(),
              "c", (np.abs(res.c - o["c"]) / np.abs(o["c"])).tolist())
        print("tol trace oracle", o["tol_trace"].tolist() if "tol_trace" in o else None)
        print("tol trace gpu   ", res.trace["tol"].tolist())
    for g in T._NUDGES:
        o2, e2 = side(lambda: oracle.fit_iht(ox, y, z * g, k=k, max_iter=40, **okw, **both))
        print("nudge", g, ("ERROR " + e2) if o2 is None else ("iter %d logl %r r %r" % (o2["iter"], o2["logl"], o2["nb_r"])))
    for t, pm in enumerate(T._row_orders(len(y), int(os.environ.get("MIH_ROW_ORDERS", 4)))):
        o2, e2 = side(lambda: oracle.fit_iht(T._rows_permuted(oracle, ox, pm), y[pm], z[pm], k=k, max_iter=40, **okw, **both))
        print("row order", t, ("ERROR " + e2) if o2 is None else ("iter %d logl %r bt %s" % (o2["iter"], o2["logl"], o2["bt_trace"].tolist())))
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
if what == "cv":
    want = int(sys.argv[3])
    fams = [("normal", "identity", mih.Normal, mih.IdentityLink, 1e-6), ("bernoulli", "logit", mih.Bernoulli, mih.LogitLink, 1e-5),
            ("poisson", "log", mih.Poisson, mih.LogLink, 1e-5)]
    for trial in range(want + 1):
        n, p, q, od, ol, D, L, tol, x, ox, y, path, folds, extra = T._cv_case(mih, oracle, rng, trial, fams)
    print("n", n, "p", p, "q", q, od, "path", path, "extra", {k: (v if not hasattr(v, "shape") else f"array{v.shape}") for k, v in extra.items()})
    mse, raw = mih.cv_iht(y, x, None, d=D(), l=L(), path=path, q=q, folds=folds, verbose=False, return_raw=True, **extra)
    omse, oraw = oracle.cv_iht(ox, y, None, path=path, q=q, folds=folds, dist=od, link=ol, **extra)
    bad = np.argwhere(~np.isclose(raw, oraw, rtol=100 * tol, atol=0))
    print("entries that differ:", bad.tolist())
    for f, j in bad[:4]:
        tr = (folds != f + 1).astype(np.uint8)
        one_o = oracle.fit_iht(ox, y, None, k=path[j], dist=od, link=ol, max_iter=100, train=tr, **extra)
        one_g = mih.fit_iht(y, x, None, k=path[j], d=D(), l=L(), max_iter=100, train=tr, verbose=False, **extra)
        print(f" fold {f + 1} k {path[j]}: cv gpu {raw[f, j]!r} oracle {oraw[f, j]!r}")
        print("   single fits: oracle iter", one_o["iter"], "logl", one_o["logl"], "bt", one_o["bt_trace"].tolist(), "| gpu iter", one_g.iter, "logl", one_g.logl, "bt", one_g.trace["backtracks"].tolist())
        print("   supports equal:", np.array_equal(np.flatnonzero(one_o["beta"]), np.flatnonzero(one_g.beta)), "max |beta diff|", float(np.max(np.abs(one_o["beta"] - one_g.beta))))
    for rr in range(2):
        again = mih.cv_iht(y, x, None, d=D(), l=L(), path=path, q=q, folds=folds, verbose=False, return_raw=True, **extra)[1]
        print(" rerun equal to the first GPU run:", np.array_equal(again, raw))
    two = [mih.cv_iht(y, x, None, d=D(), l=L(), path=path, q=q, folds=folds, verbose=False, return_raw=True, rank=r, world=2, **extra)[1] for r in range(2)]
    print(" two shards add up to the single-rank matrix:", np.array_equal(two[0] + two[1], raw))
    if extra.get("init_beta") and len(bad):
        f, j = bad[0]
        tr = (folds != f + 1).astype(np.uint8)
        for kk in path[:3]:
            for mi in (2, 3):
                o = oracle.fit_iht(ox, y, None, k=kk, dist=od, link=ol, max_iter=mi, train=tr, init_beta=True)
                g = mih.fit_iht(y, x, None, k=kk, d=D(), l=L(), max_iter=mi, train=tr, verbose=False, init_beta=True)
                so, sg = np.flatnonzero(o["beta"]), np.flatnonzero(g.beta)
                print(f"  k {kk} max_iter {mi}: oracle support {so.tolist()} beta {o['beta'][so].round(6).tolist()} c {o['c'].round(6).tolist()} logl {o['logl']:.6f} | gpu support {sg.tolist()} beta {g.beta[sg].round(6).tolist()} c {np.round(g.c, 6).tolist()} logl {g.logl:.6f}")
        # the univariate regressions themselves on the training rows (numpy)
        cols = x.export_bed()
        mu, sv = x.mu_sigma()
        tri = np.flatnonzero(tr)
        ys = y[tri]
        b1 = np.zeros(p); b0 = np.zeros(p); fail = 0
        for jj in range(p):
            code = np.stack([(cols[jj] >> (2 * t)) & 3 for t in range(4)], axis=1).ravel()[:n]
            gdos = np.array([0.0, np.nan, 1.0, 2.0])[code]
            gdos[np.isnan(gdos)] = mu[jj]
            xs = ((gdos - mu[jj]) * sv[jj])[tri]
            N = xs.size; sx = xs.sum(); sxx = (xs * xs).sum(); sxy = xs @ ys; sy = ys.sum()
            d = sxx - sx * sx / N
            if not d > 0: fail += 1; b0[jj], b1[jj] = sy, sxy
            else: b1[jj] = (sxy - sx * sy / N) / d; b0[jj] = (sy - b1[jj] * sx) / N
        cl = np.clip(b1, -2, 2)
        print("  numpy regressions: failed Cholesky", fail, "clamped to +-2:", int(np.sum(np.abs(cl) == 2.0)), "top |beta|:", np.sort(np.abs(cl))[::-1][:6].round(6).tolist(),
              "argsort top:", np.argsort(-np.abs(cl), kind="stable")[:6].tolist())
if what == "mvcv":
    want = int(sys.argv[3])
    for trial in range(want + 1):
        n, p, r, qz, q, x, ox, Y, Z, path, folds, extra = T._mvcv_case(mih, oracle, rng, trial)
    print("n", n, "p", p, "r", r, "qz", qz, "q", q, "path", path, "extra", extra)
    mse, raw = mih.cv_iht(Y, x, Z, path=path, q=q, folds=folds, verbose=False, return_raw=True, **extra)
    omse, oraw = oracle.cv_mv(ox, Y, Z, path=path, q=q, folds=folds, **extra)
    print("gpu", raw.tolist()); print("oracle", oraw.tolist())
    if extra.get("init_beta"):
        for f in range(q):
            tr = (folds != f + 1).astype(np.uint8)
            one = oracle.fit_iht(ox, Y[0], None, k=path[0], max_iter=2, train=tr, init_beta=True)
            print(" fold", f + 1, "ib_cond of the univariate regressions on its training rows:", one["ib_cond"])
if what == "paths":
    want = int(sys.argv[3])
    fams = [("normal", "identity", mih.Normal, mih.IdentityLink), ("bernoulli", "logit", mih.Bernoulli, mih.LogitLink),
            ("poisson", "log", mih.Poisson, mih.LogLink), ("negbin", "log", mih.NegativeBinomial, mih.LogLink)]
    for trial in range(want + 1):
        n, p, q, od, ol, D, L, x, ox, y, z, path, kw, okw, d = T._path_case(mih, oracle, rng, trial, fams)
    print("n", n, "p", p, "q", q, od, "path", path, "kw", {k: (v if not hasattr(v, "shape") else "array") for k, v in kw.items()}, "d", d)
    ll = np.asarray(mih.iht_run_many_models(y, x, z, path=path, d=d, l=L(), verbose=False, **kw))
    for k, got in zip(path, ll):
        o = oracle.fit_iht(ox, y, z, k=k, dist=od, link=ol, max_iter=100, **okw)
        g1 = mih.fit_iht(y, x, z, k=k, d=d, l=L(), max_iter=100, verbose=False, **kw)
        print(f" k {k}: path {got!r} single gpu fit {g1.logl!r} (iter {g1.iter}, r {getattr(g1.d, 'r', None)}) oracle {o['logl']!r} (iter {o['iter']}, r {o['nb_r']}, bt max {o['bt_trace'].max(initial=0)})")
        for g in T._NUDGES[:3]:
            o2 = oracle.fit_iht(ox, y, z * g, k=k, dist=od, link=ol, max_iter=100, **okw)
            print(f"     nudge {g}: oracle logl {o2['logl']!r} iter {o2['iter']} r {o2['nb_r']}")
if what == "fits":
    want = int(sys.argv[3])
    fams = [("normal", "identity", mih.Normal, mih.IdentityLink, 1e-5), ("bernoulli", "logit", mih.Bernoulli, mih.LogitLink, 1e-4),
            ("poisson", "log", mih.Poisson, mih.LogLink, 1e-4)]
    for trial in range(want + 1):
        n, p, k, miss, q, od, ol, D, L, tol, x, ox, y, z, kw = T._fits_case(mih, oracle, rng, trial, fams)
    print("n", n, "p", p, "k", k, od, "q", q, miss, sorted(kw))
    o = oracle.fit_iht(ox, y, z, k=k, dist=od, link=ol, max_iter=60, **kw)
    res = mih.fit_iht(y, x, z, k=k, d=D(), l=L(), max_iter=60, verbose=False, **kw)
    nz = np.flatnonzero(o["beta"])
    print("iter", res.iter, o["iter"], "logl", res.logl, o["logl"], "bt", o["bt_trace"].tolist())
    print("beta gpu", res.beta[nz].tolist(), "\nbeta orc", o["beta"][nz].tolist(), "\n diff", (res.beta[nz] - o["beta"][nz]).tolist())
    print("c gpu", res.c.tolist(), "orc", o["c"].tolist(), "diff", (res.c - o["c"]).tolist())
    for g in T._NUDGES:
        o2 = oracle.fit_iht(ox, y, z * g, k=k, dist=od, link=ol, max_iter=60, **kw)
        print(f"  nudge {g!r}: iter {o2['iter']} beta diff {(o2['beta'][nz] - o['beta'][nz]).tolist()} c diff {(o2['c'] * g - o['c']).tolist()}")
if what == "mvfits":
    want = int(sys.argv[3])
    for trial in range(want + 1):
        n, p, r, q, k, miss, x, ox, Y, Z, kw = T._mvfit_case(mih, oracle, rng, trial)
    print("n", n, "p", p, "r", r, "q", q, "k", k, miss, {a: (b if not hasattr(b, "shape") else "array") for a, b in kw.items()})
    o = oracle.fit_mv(ox, Y, Z, k=k, max_iter=60, **kw)
    res = mih.fit_iht(Y, x, Z, k=k, max_iter=60, verbose=False, **kw)
    print("iter", res.iter, o["iter"], "bt gpu", res.trace["backtracks"].tolist(), "bt orc", o["bt_trace"].tolist())
    m_ = min(len(res.trace["logl"]), len(o["logl_trace"]))
    for i in range(m_):
        print(f"  {i}: logl {res.trace['logl'][i]!r} {o['logl_trace'][i]!r} tol {res.trace['tol'][i]!r} {o['tol_trace'][i]!r}")
    for extra in (res.trace["tol"][m_:], o["tol_trace"][m_:]):
        print("  further tol", list(extra))
    if kw.get("init_beta"):
        one = oracle.fit_iht(ox, Y[0], None, k=1, max_iter=1, train=kw.get("train"), init_beta=True)
        print("  ib_cond of the univariate regressions on the training rows:", one["ib_cond"])
    for g in T._NUDGES:
        o2 = oracle.fit_mv(ox, Y, Z * g, k=k, max_iter=60, **kw)
        print(f"  nudge {g!r}: iter {o2['iter']} last tol {o2['tol_trace'][-1]!r} bt {o2['bt_trace'].tolist()}")
