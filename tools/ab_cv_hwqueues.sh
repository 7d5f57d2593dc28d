#!/bin/bash
# tools/ab_cv_hwqueues.sh -- cv_iht at configs[3] size with more hardware queues behind the HIP streams (GPU_MAX_HW_QUEUES, default 4: the
# 19 fits' streams and the lanes' two share four AQL queues, and packets of one queue start in order).  Separate processes, one box.
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" python - "$name" <<'PY'
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
m.set_step_mode(int(os.environ.get("MIH_MODE", "0")))
ts = []
for rep in range(4):
    t0 = time.perf_counter()
    mse, raw = m.cv_iht(y, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink())
    ts.append(round(time.perf_counter() - t0, 3))
print(json.dumps({"variant": sys.argv[1], "seconds": ts[1:], "hash": hashlib.sha256(raw.tobytes()).hexdigest()[:12]}), flush=True)
PY
}
for rep in 1 2; do
run "resident per fit, 4 hardware queues (default)" MIH_MODE=0
run "resident per fit, 8 hardware queues" MIH_MODE=0 GPU_MAX_HW_QUEUES=8
run "resident per fit, 16 hardware queues" MIH_MODE=0 GPU_MAX_HW_QUEUES=16
run "resident per fit, 24 hardware queues" MIH_MODE=0 GPU_MAX_HW_QUEUES=24
run "host-driven, 16 hardware queues" MIH_MODE=1 GPU_MAX_HW_QUEUES=16
run "resident per fit, 2 hardware queues" MIH_MODE=0 GPU_MAX_HW_QUEUES=2
done
