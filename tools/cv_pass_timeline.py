#!/usr/bin/env python3
"""tools/cv_pass_timeline.py [step_mode] -- the fused passes of one configs[3] cross-validation as a timeline: per pass its lane, start,
duration, residuals and the gap to the end of the pass before it (any lane).  Where the time without a pass in flight sits."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mendeliht_amd as m
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, p = 500_000, int(os.environ.get("MIH_P", 1_000_000))
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = m.hash_folds(n, 5)
m.set_step_mode(mode)
kw = dict(path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink())
m.cv_iht(y, x, None, **kw)
m.profile_read(x, reset=True); m.profile_enable(x, True)
t0 = time.perf_counter()
m.cv_iht(y, x, None, **kw)
wall = time.perf_counter() - t0
m.profile_enable(x, False)
ps = sorted(m.profile_passes(x, reset=True), key=lambda q: q["start_ms"])
t_first = ps[0]["start_ms"]
end = t_first
gaps = []
for i, q in enumerate(ps):
    gap = q["start_ms"] - end
    gaps.append(max(gap, 0.0))
    if i < 30 or i >= len(ps) - 8:
        print(f"pass {i:2d} lane {q['stream_tag']} start {q['start_ms'] - t_first:8.2f} ms  dur {q['ms']:6.2f}  residuals {q['residuals']:2d}  gap {gap:7.2f}")
    end = max(end, q["start_ms"] + q["ms"])
print(json.dumps({"step_mode": mode, "wall_s": round(wall, 3), "passes": len(ps), "first_to_last_ms": round(end - t_first, 1), "pass_ms_sum": round(sum(q["ms"] for q in ps), 1),
                  "gap_ms_sum": round(sum(gaps), 1), "before_first_pass_and_after_last_ms": round(1e3 * wall - (end - t_first), 1)}))
