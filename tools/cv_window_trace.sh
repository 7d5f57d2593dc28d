#!/bin/bash
# tools/cv_window_trace.sh MODE -- rocprofv3 kernel trace of one configs[3] cross-validation (step_mode MODE); the analysis lists what runs
# in the windows between two fused passes.
mode=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/cvtrace_$mode
cat > /tmp/cv_once.py <<PY
import os, sys
sys.path.insert(0, "$R")
import numpy as np
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = m.hash_folds(n, 5)
m.set_step_mode($mode)
m.cv_iht(y, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink())
PY
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cvtrace_$mode -o t -- python3 /tmp/cv_once.py > /dev/null 2> $R/gpurun_out/cvtrace_$mode.err
f=$(find $R/gpurun_out/cvtrace_$mode -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "")) for r in rows]
ks.sort()
passes = [k for k in ks if "k_xtv_dma16" in k[2]]
print("kernels", len(ks), "passes", len(passes))
# windows between consecutive passes in the steady part
tot = collections.Counter(); cnt = collections.Counter(); wins = []
for a, b in zip(passes[10:40], passes[11:41]):
    lo, hi = a[1], b[0]
    if hi <= lo: continue
    inside = [k for k in ks if k[0] >= lo - 2_000_000 and k[1] <= hi + 1000 and "k_xtv_dma16" not in k[2]]
    busy = sum(k[1] - k[0] for k in inside if k[0] >= lo)
    wins.append(((hi - lo) / 1e6, len([k for k in inside if k[0] >= lo]), busy / 1e6))
    for k in inside:
        if k[0] >= lo: tot[k[2]] += k[1] - k[0]; cnt[k[2]] += 1
print("windows (ms, kernels, summed kernel ms):", [(round(w, 2), c, round(b, 2)) for w, c, b in wins])
for name, t in tot.most_common(25):
    print(f"{t / 1e6:9.2f} ms  {cnt[name]:6d} x  {t / cnt[name] / 1e3:8.1f} us  {name}")
# how many kernels ran DURING passes (started and ended inside one)
during = 0
for k in ks:
    if "k_xtv_dma16" in k[2]: continue
    for ps in passes:
        if k[0] > ps[0] and k[1] < ps[1]: during += 1; break
print("non-pass kernels that ran entirely inside a pass:", during)
PY
