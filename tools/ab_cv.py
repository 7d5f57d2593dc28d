"""A/B of two builds of the library on cv_iht at BASELINE configs[3] size (100 Bernoulli/Logit fits, n = 500k, p = 1M): alternating
processes, one warm-up and three timed cross-validations each, the 5 x 20 losses bit for bit.  usage: ab_cv.py OLD.so NEW.so [rounds]"""
import json, os, subprocess, sys
old, new = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
snippet = r'''
import hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, %r)
import mendeliht_amd as m
n, p = 500_000, int(os.environ.get("MIH_P", 1_000_000))
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = m.hash_folds(n, 5)
ts = []
for rep in range(4):
    t0 = time.perf_counter()
    mse, raw = m.cv_iht(y, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink())
    ts.append(time.perf_counter() - t0)
print(json.dumps(dict(seconds=[round(t, 3) for t in ts[1:]], best_k=int(np.argmin(mse)) + 1, hash=hashlib.sha256(raw.tobytes()).hexdigest()[:16])), flush=True)
''' % ROOT
res = []
for rnd in range(rounds):
    for name, lib in (("old", old), ("new", new)):
        r = subprocess.run([sys.executable, "-c", snippet], env=dict(os.environ, MENDELIHT_HIP_LIB=lib), capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
        if not line.startswith("{"):
            print(name, "FAILED", r.stderr[-1500:]); sys.exit(1)
        res.append((name, json.loads(line)))
        print(rnd, name, line, flush=True)
same = len({d["hash"] for _, d in res}) == 1
print("losses bit-identical:", same, " min seconds: old", min(min(d["seconds"]) for n_, d in res if n_ == "old"),
      "new", min(min(d["seconds"]) for n_, d in res if n_ == "new"))
sys.exit(0 if same else 1)
