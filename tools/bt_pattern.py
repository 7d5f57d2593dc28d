"""The backtrack count of every step of the bench's fit (configs[2], bench.py's data), one iht_one_step! per call.
usage: python tools/bt_pattern.py [steps]   -- what the attempt-slot forecast of the resident chain (fit.hip: res_spec) has to predict"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import mendeliht_amd as m
n, p, k = 500000, int(os.environ.get("MIH_BENCH_P", 1000000)), 200
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 105
x = m.SnpLinAlg.synthetic(n, p, seed=2024, device=0)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, size=k, replace=False)); beta = rng.standard_normal(k)
y = x.xv_sparse(supp, beta) + 1.0 + rng.standard_normal(n)
s = m.IHTSession(y, x, None, k=k, d=m.Normal(), l=m.IdentityLink())
seq = []
for _ in range(steps):
    _l, nbt, _t = s.step()
    seq.append(int(nbt))
print("backtracks per step:", "".join(str(min(b, 9)) for b in seq))
print("counts:", {b: seq.count(b) for b in sorted(set(seq))})
