#!/bin/bash
# tools/ab_cv_share.sh -- one GPU's share of the configs[3] cross-validation at world = 8 (rank 0: 13 fits, ONE lane), measurement build,
# separate processes: host-driven steps, resident with one chain per fit, resident batched over the lane.
cd $GRAFT_REPO_ROOT
run() {
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
out = {}
for rank in (0, 3):
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        mse, raw = m.cv_iht(y, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink(), rank=rank, world=8)
        ts.append(round(time.perf_counter() - t0, 3))
    out[f"rank{rank}"] = ts[1:]
    out[f"hash{rank}"] = hashlib.sha256(raw.tobytes()).hexdigest()[:10]
print(json.dumps({"variant": sys.argv[1], **out}), flush=True)
PY
}
run "host-driven (round 5)" MIH_MODE=1
run "resident, one chain per fit" MIH_MODE=0 MENDELIHT_LANE_PER_FIT=1
run "resident, batched over the lane" MIH_MODE=0
