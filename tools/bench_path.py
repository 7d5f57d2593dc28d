"""iht_run_many_models (full-data fits for path=1:20) at n=500k, p=1M: lock-step fused passes vs one fit at a time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
n, p = 500_000, int(os.environ.get("MIH_P", 1_000_000))
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, size=10, replace=False))
y = x.xv_sparse(supp, rng.standard_normal(10)) + 1.0 + rng.standard_normal(n)
path = list(range(1, 21))
for digits in (0, 4908):
    m.set_xtv_digits(digits)
    t0 = time.perf_counter(); ll = m.iht_run_many_models(y, x, None, path=path, verbose=False); dt = time.perf_counter() - t0
    print(f"digits={digits}: lock-step path=1:20 in {dt:.2f} s (argmax of the logl increments at k={int(np.argmax(np.diff(ll) < 1.0)) + 1})", flush=True)
m.set_xtv_digits(0)
t0 = time.perf_counter()
seq = [m.fit_iht(y, x, None, k=k, verbose=False, max_iter=100).logl for k in path]
dt = time.perf_counter() - t0
print(f"one fit at a time: {dt:.2f} s; max |difference| of the loglikelihoods {np.max(np.abs(np.array(seq) - m.iht_run_many_models(y, x, None, path=path, verbose=False))):.3g}")
