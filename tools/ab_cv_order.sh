#!/bin/bash
# tools/ab_cv_order.sh -- cv_iht at configs[3] size on the MEASUREMENT build: in which order the queue hands out fits whose length is not
# known yet (CvQueue: the caller's fold-major order, or the larger / smaller model sizes first).  Same results whatever the order.
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env MENDELIHT_HIP_PROBES=1 "$@" python - "$name" <<'PY'
import os, sys, time, hashlib, json
sys.path.insert(0, os.getcwd())
import numpy as np
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = m.hash_folds(n, 5)
ts = []
for rep in range(4):
    m.profile_read(x, reset=True); m.profile_enable(x, True)
    t0 = time.perf_counter()
    mse, raw = m.cv_iht(y, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink())
    ts.append(round(time.perf_counter() - t0, 3))
    m.profile_enable(x, False)
    ps = m.profile_passes(x, reset=True)
print(json.dumps({"variant": sys.argv[1], "seconds": ts[1:], "passes": len(ps), "residuals": sum(q["residuals"] for q in ps), "hash": hashlib.sha256(raw.tobytes()).hexdigest()[:12]}), flush=True)
PY
}
for rep in 1 2; do
run "caller's order (fold-major)"
run "larger model sizes first" MENDELIHT_CV_ORDER=kdesc
run "smaller model sizes first" MENDELIHT_CV_ORDER=kasc
done
