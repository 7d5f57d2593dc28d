#!/usr/bin/env python3
"""tools/ab_steps_libs.py OLD.so NEW.so -- configs[2] session steps with two builds of the library, alternating processes: resident steps
with and without the measurement hook, host-driven steps; ms per step and ms outside the pass."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
snippet = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np
import mendeliht_amd as m
n, p, k = 500_000, 1_000_000, 200
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, size=k, replace=False))
y = x.xv_sparse(supp, rng.standard_normal(k)) + 1.0 + rng.standard_normal(n)
out = {}
def run(mode, hook, steps, tag):
    s = m.IHTSession(y, x, None, k=k, step_mode=mode)
    for _ in range(5): s.step()
    m.profile_read(x, reset=True)
    if hook: m.profile_enable(x, True)
    t0 = time.perf_counter(); s.run(steps); dt = 1e3 * (time.perf_counter() - t0) / steps
    kern = None
    if hook:
        m.profile_enable(x, False)
        ms, cnt = m.profile_read(x, reset=True)
        kern = ms / max(cnt, 1)
    s.close()
    out[tag] = {"ms_per_step": round(dt, 4), "kernel_ms": None if kern is None else round(kern, 4), "outside": None if kern is None else round(dt - kern, 4)}
run(0, True, 60, "resident_hooked")
run(1, True, 30, "host_hooked")
run(0, False, 60, "resident_unhooked")
run(1, False, 30, "host_unhooked")
run(0, True, 60, "resident_hooked_again")
print(json.dumps(out))
''' % ROOT
for rnd in range(2):
    for name, lib in (("old", sys.argv[1]), ("new", sys.argv[2])):
        r = subprocess.run([sys.executable, "-c", snippet], env=dict(os.environ, MENDELIHT_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:]
        print(rnd, name, line, flush=True)
