"""Wall time of the bench's fit per iht_one_step! with the measurement hook OFF and ON (the hook brackets every X'r pass with two HIP
event records, ~5.7 us each, which land in bench.py's "outside the pass" figure).  usage: python tools/step_wall.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mendeliht_amd as m
n, p, k = 500000, int(os.environ.get("MIH_BENCH_P", 1000000)), 200
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
x = m.SnpLinAlg.synthetic(n, p, seed=2024, device=0)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, size=k, replace=False)); beta = rng.standard_normal(k)
y = x.xv_sparse(supp, beta) + 1.0 + rng.standard_normal(n)
for hook in (False, True, False, True):
    s = m.IHTSession(y, x, None, k=k, d=m.Normal(), l=m.IdentityLink())
    for _ in range(5):
        s.step()
    m.profile_read(x, reset=True)
    m.profile_enable(x, hook)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.run(steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    m.profile_enable(x, False)
    ps = m.profile_passes(x, reset=True) if hook else []
    kern = sum(q["ms"] for q in ps) / max(len(ps), 1) if ps else float("nan")
    print(f"hook {'on ' if hook else 'off'}: {1e3 * dt / steps:.4f} ms per step" + (f", pass kernel {kern:.4f} ms, outside {1e3 * dt / steps - kern:.4f} ms" if hook else ""))
    s.close()
