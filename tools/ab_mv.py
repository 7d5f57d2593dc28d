"""A/B of two builds of the library on the multivariate path: BASELINE configs[4] (r = 10 traits, k = 500, n = 500k, p = 1M), a small fit
that backtracks, and a small multivariate cross-validation -- every B / C / Sigma / loglikelihood trace must be bit-identical
between the two builds; prints ms per iteration of configs[4] for each.  usage: ab_mv.py OLD.so NEW.so"""
import hashlib, json, os, subprocess, sys
old, new = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
snippet = r'''
import hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, %r)
import mendeliht_amd as m
def h(*arrs):
    d = hashlib.sha256()
    for a in arrs: d.update(np.ascontiguousarray(a).tobytes())
    return d.hexdigest()[:16]
out = {}
n, p, r, k = 500_000, int(os.environ.get("MIH_P", 1_000_000)), 10, 500
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(3)
lin = rng.choice(r * p, k, replace=False)
Y = rng.standard_normal((r, n))
for t in range(r):
    cols = np.unique(lin[lin %% r == t] // r)
    Y[t] += x.xv_sparse(cols, rng.standard_normal(cols.size) * 0.3) + 1.0
m.fit_iht(Y, x, None, k=k, verbose=False, max_iter=3)
m.profile_read(x, reset=True); m.profile_enable(x, True)
res = m.fit_iht(Y, x, None, k=k, verbose=False, max_iter=12)
m.profile_enable(x, False)
passes = m.profile_passes(x, reset=True)
kern = sum(q["ms"] for q in passes[1:])                  # (the first launch is the initial score, outside res.time)
steps = len(passes) - 1
gaps = [passes[i + 1]["start_ms"] - passes[i]["start_ms"] - passes[i]["ms"] for i in range(1, len(passes) - 1)]      # end of a step's pass -> start of the next step's
out["config4"] = dict(ms_per_iteration=1e3 * res.time / steps, pass_kernel_ms=kern / steps, outside_the_pass_kernel_ms=(1e3 * res.time - kern) / steps,
                      pass_to_pass_ms=sum(gaps) / max(len(gaps), 1),
                      iters=int(res.iter), hash=h(res.beta, res.c, res.Sigma, res.trace["logl"], res.trace["tol"]),
                      backtracks=int(res.trace["backtracks"].sum()))
del x
xs = m.SnpLinAlg.synthetic(3000, 800, seed=5, missing_rate=0.02)
rng = np.random.default_rng(9)
Z = np.vstack([np.ones(3000), rng.standard_normal(3000), rng.standard_normal(3000)])
Ys = rng.standard_normal((4, 3000)) * 2
for t in range(4):
    cc = rng.choice(800, 3, replace=False)
    Ys[t] += xs.xv_sparse(np.sort(cc), rng.standard_normal(3)) + 0.5 * Z[1]
for kw in (dict(k=7), dict(k=12, zkeep=[1, 0, 1]), dict(k=9, init_beta=True)):
    rr = m.fit_iht(Ys, xs, Z, verbose=False, **kw)
    out[str(sorted(kw.items()))] = dict(hash=h(rr.beta, rr.c, rr.Sigma, rr.trace["logl"], rr.trace["tol"]), iters=int(rr.iter), bt=int(rr.trace["backtracks"].sum()))
folds = m.hash_folds(3000, 3)
mse, raw = m.cv_iht(Ys, xs, Z, path=[2, 5, 8, 11], q=3, folds=folds, verbose=False, return_raw=True)
out["cv"] = h(raw)
print(json.dumps(out), flush=True)
''' % ROOT
res = {}
for name, lib in (("old", old), ("new", new), ("old2", old), ("new2", new)):
    env = dict(os.environ, MENDELIHT_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-c", snippet], env=env, capture_output=True, text=True)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    if not line.startswith("{"):
        print(name, "FAILED", r.stderr[-2000:]); sys.exit(1)
    res[name] = json.loads(line)
    print(name, line, flush=True)
strip = lambda d: {k: ({kk: vv for kk, vv in v.items() if not kk.endswith("_ms") and kk != "ms_per_iteration"} if isinstance(v, dict) else v) for k, v in d.items()}
same = strip(res["old"]) == strip(res["new"]) == strip(res["new2"])
print("bit-identical:", same, " config4 ms outside the pass kernel (finalize, statistics and digit planes included): old",
      [round(res[k]["config4"]["outside_the_pass_kernel_ms"], 3) for k in ("old", "old2")],
      "new", [round(res[k]["config4"]["outside_the_pass_kernel_ms"], 3) for k in ("new", "new2")],
      " from the end of a step's pass to the start of the next step's (HIP events, no end effects): old",
      [round(res[k]["config4"]["pass_to_pass_ms"], 3) for k in ("old", "old2")], "new", [round(res[k]["config4"]["pass_to_pass_ms"], 3) for k in ("new", "new2")])
sys.exit(0 if same else 1)
