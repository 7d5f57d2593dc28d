"""tools/coexist_probe.py -- bench.py's sequence around its timed region in small: timed resident steps, host-driven steps of a second
session beside the idle resident one, the resident session continued without the measurement hook; with / without torch in the process."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
import mendeliht_amd as m
def sync():
    if os.environ.get("WITH_TORCH"): torch.cuda.synchronize()
n, p, k = 500_000, 1_000_000, 200
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, size=k, replace=False))
y = x.xv_sparse(supp, rng.standard_normal(k)) + 1.0 + rng.standard_normal(n)
A = m.IHTSession(y, x, None, k=k, step_mode=0)
for _ in range(5): A.step()
m.profile_read(x, reset=True); m.profile_counters(x, reset=True); m.profile_exchange(x, reset=True); m.profile_enable(x, True)
sync(); t0 = time.perf_counter(); A.run(100); sync(); dt = 1e3 * (time.perf_counter() - t0) / 100
m.profile_enable(x, False)
ps = m.profile_passes(x, reset=True); kern = sum(q["ms"] for q in ps) / len(ps)
print("timed resident: ms/step", round(dt, 3), "kernel", round(kern, 3), "outside", round(dt - kern, 3), flush=True)
s2 = m.IHTSession(y, x, None, k=k, step_mode=1)
for _ in range(5): s2.step()
m.profile_read(x, reset=True); m.profile_enable(x, True)
sync(); t0 = time.perf_counter(); s2.run(20); sync(); dt2 = 1e3 * (time.perf_counter() - t0) / 20
m.profile_enable(x, False)
ps2 = m.profile_passes(x, reset=True); k2 = sum(q["ms"] for q in ps2) / len(ps2)
print("host-driven beside it: ms/step", round(dt2, 3), "kernel", round(k2, 3), "outside", round(dt2 - k2, 3), "launches", len(ps2), flush=True)
s2.close(); del s2
sync(); t0 = time.perf_counter(); A.run(20); sync(); dt3 = 1e3 * (time.perf_counter() - t0) / 20
print("continued unhooked: ms/step", round(dt3, 3), "minus the timed region's kernel", round(dt3 - kern, 3), flush=True)
m.profile_read(x, reset=True); m.profile_enable(x, True)
sync(); t0 = time.perf_counter(); A.run(20); sync(); dt4 = 1e3 * (time.perf_counter() - t0) / 20
m.profile_enable(x, False)
ps4 = m.profile_passes(x, reset=True); k4 = sum(q["ms"] for q in ps4) / len(ps4)
print("continued hooked: ms/step", round(dt4, 3), "kernel", round(k4, 3), "outside", round(dt4 - k4, 3), flush=True)
