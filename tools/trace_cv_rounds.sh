cd $GRAFT_REPO_ROOT
for mode in 1 0; do
MIH_MODE=$mode MENDELIHT_HIP_PROBES=1 MENDELIHT_CV_TRACE=1 python - <<'PY' 2> gpurun_out/cv_trace_mode$mode.txt
import os, sys, time
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
m.set_step_mode(int(os.environ["MIH_MODE"]))
for rep in range(2):
    sys.stderr.write(f"=== rep {rep}\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    m.cv_iht(y, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink())
    sys.stderr.write(f"=== seconds {time.perf_counter() - t0:.3f}\n")
PY
done
python - <<'PY'
import re
for mode in (1, 0):
    txt = open(f"gpurun_out/cv_trace_mode{mode}.txt").read().split("=== rep 1")[1]
    rows = re.findall(r"lane (\d) round (\d+): ([\d.]+) ms \(before the pass ([\d.]+) ms, behind it ([\d.]+) ms of host time\), (\d+) scores", txt)
    tot = [0, 0]; pre = [0, 0]; post = [0, 0]; cnt = [0, 0]
    for l, r, t, a, b, s in rows:
        l = int(l); tot[l] += float(t); pre[l] += float(a); post[l] += float(b); cnt[l] += 1
    print("mode", mode, "rounds", cnt, "round ms sum", [round(v) for v in tot], "pre", [round(v) for v in pre], "post", [round(v) for v in post], re.findall(r"=== seconds ([\d.]+)", txt))
    for row in rows[:12]: print("   ", row)
PY
