#!/bin/bash
# tools/ab_cv_lanes.sh -- cv_iht at configs[3] size on the MEASUREMENT build, separate processes on one box: the lanes' fits resident
# (step_mode 0) or host-driven (1), with / without the single-file order of the lanes' passes and the worker streams' priority.
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
run "host-driven, no CUs reserved" MIH_MODE=1 MENDELIHT_LANE_CU_RESERVE=0
run "host-driven, 8 CUs reserved" MIH_MODE=1 MENDELIHT_LANE_CU_RESERVE=8
run "resident per fit, no CUs reserved" MIH_MODE=0 MENDELIHT_LANE_CU_RESERVE=0
run "resident per fit, 8 CUs reserved" MIH_MODE=0 MENDELIHT_LANE_CU_RESERVE=8
run "resident per fit, 16 CUs reserved" MIH_MODE=0 MENDELIHT_LANE_CU_RESERVE=16
run "resident per fit, 32 CUs reserved" MIH_MODE=0 MENDELIHT_LANE_CU_RESERVE=32
run "resident batched, 16 CUs reserved" MIH_MODE=0 MENDELIHT_LANE_CU_RESERVE=16 MENDELIHT_LANE_BATCHED=1
done
