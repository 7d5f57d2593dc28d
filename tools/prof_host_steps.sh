#!/bin/bash
# tools/prof_host_steps.sh MODE -- rocprofv3 kernel statistics of 30 session steps (configs[2]) with step_mode MODE
mode=${1:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/steps.py <<PY
import sys, time
sys.path.insert(0, "$R")
import numpy as np
import mendeliht_amd as m
n, p, k = 500_000, 1_000_000, 200
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, size=k, replace=False))
y = x.xv_sparse(supp, rng.standard_normal(k)) + 1.0 + rng.standard_normal(n)
s = m.IHTSession(y, x, None, k=k, step_mode=$mode)
for _ in range(5): s.step()
t0 = time.perf_counter(); s.run(30); print("ms per step", 1e3 * (time.perf_counter() - t0) / 30)
PY
rm -rf $R/gpurun_out/hoststeps_$mode
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hoststeps_$mode -o t -- python3 /tmp/steps.py
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/hoststeps_$mode/t_kernel_stats.csv")))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:9.1f} total_ms {float(r['TotalDurationNs']) / 1e6:8.2f}")
PY
