"""Race hunt for the lock-step drivers (coroutines of the lane thread, per-fit worker streams, events around the fused pass, tail
hand-over): the same cross-validations and model paths over and over, from several host threads at once, every result compared bit
for bit with the first."""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd.api as _api
_api.RESERVE_BY_DEFAULT = True          # small matrices with the reserve a 125 GB matrix keeps (mih_mat_reserve)
import mendeliht_amd as m
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = []
for (n, p, q, path, seed) in ((6001, 900, 3, range(1, 9), 3), (20_000, 4000, 5, range(1, 21), 5), (3000, 40_000, 4, [2, 5, 9, 14, 20, 30, 40], 7)):
    x = m.SnpLinAlg.synthetic(n, p, seed=seed, missing_rate=0.01 if seed == 3 else 0.0)
    rng = np.random.default_rng(seed)
    supp = np.sort(rng.choice(p, 10, replace=False))
    eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    y = eta + rng.standard_normal(n)
    folds = m.hash_folds(n, q)
    ynb = rng.negative_binomial(4, 4 / (np.exp(0.5 + 0.3 * eta) + 4)).astype(float)      # (round 4: est_r chains on the lock-step driver)
    cases.append((x, yb, y, folds, q, list(path), ynb))


def run_all():
    out = []
    for x, yb, y, folds, q, path, ynb in cases:
        nb = dict(d=m.NegativeBinomial(1.0), l=m.LogLink(), verbose=False)
        if x.p <= 4000:                                        # chains of NegBin fits handing r on, in lock-step (every chain shape), and est_r model paths
            for T, est in ((0, "Newton"), (1, "MM"), (7, "Newton")):
                out.append(m.cv_iht(ynb, x, None, path=path[:6], q=q, folds=folds, return_raw=True, est_r=est, cv_threads=T, **nb)[1])
            out.append(np.asarray(m.iht_run_many_models(ynb, x, None, path=path[:6], est_r="Newton", **nb)))
        out.append(m.cv_iht(yb, x, None, path=path, q=q, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink())[1])
        out.append(m.cv_iht(y, x, None, path=path, q=q, folds=folds, verbose=False, return_raw=True)[1])
        out.append(np.asarray(m.iht_run_many_models(y, x, None, path=path, verbose=False)))
        for r in range(3):
            out.append(m.cv_iht(yb, x, None, path=path, q=q, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink(), rank=r, world=3)[1])
    return out


ref = run_all()
bad = 0
for rep in range(reps):
    res = [None, None, None]

    def work(i):
        res[i] = run_all()
    th = [threading.Thread(target=work, args=(i,)) for i in range(3 if rep % 2 else 1)]
    for t in th: t.start()
    for t in th: t.join()
    for got in res:
        if got is None:
            continue
        for a, b in zip(got, ref):
            if not np.array_equal(a.view(np.uint64), b.view(np.uint64)):
                bad += 1
    print(f"rep {rep}: {'ok' if not bad else str(bad) + ' MISMATCHES'}", flush=True)
print("stress:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
