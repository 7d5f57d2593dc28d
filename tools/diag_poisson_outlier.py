#!/usr/bin/env python3
"""tools/diag_poisson_outlier.py -- the Poisson fit with a planted count outlier (tests/test_gpu_parity.py) iteration by iteration
against the oracle: relative difference of the loglikelihood trace, how often the outlier guard fired (csrc/peel.h)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mendeliht_amd as mih
from oracle import oracle as O

n = 1000
bed = mih.read_bed(os.path.join(ROOT, "tests", "fixtures", "normal.bed"), n)
x = mih.SnpLinAlg(bed, n, center=True, scale=True, impute=True)
ox = O.Mat.from_bed_columns(bed, n)
rng = np.random.default_rng(77)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as T
eta = T._sim(O, ox, rng, 6, scale=0.25)
y = rng.poisson(np.exp(eta)).astype(float)
i0 = int(np.argmin(np.abs(eta)))
y[i0] = 500.0
mih.profile_enable(x, True)
for k in (6, 10):
    mih.profile_counters(x, reset=True)
    res = mih.fit_iht(y, x, None, k=k, d=mih.Poisson(), l=mih.LogLink(), verbose=False)
    c = mih.profile_counters(x, reset=True)
    o = O.fit_iht(ox, y, None, k=k, dist="poisson", link="log")
    ll, ol = np.asarray(res.trace["logl"]), np.asarray(o["logl_trace"])
    m = min(len(ll), len(ol))
    rel = np.abs(ll[:m] - ol[:m]) / np.abs(ol[:m])
    print(f"k={k}: iter {res.iter} / {o['iter']}, peeled {c['peeled_residuals']}, steps {len(ll)}")
    for i in range(0, m, max(1, m // 40)):
        print(f"  it {i + 1:3d} logl {ol[i]: .10e} rel diff {rel[i]:.2e} bt {res.trace['backtracks'][i]}")
    nz = np.flatnonzero(o["beta"])
    print("  beta rel diff", np.abs(res.beta[nz] - o["beta"][nz]) / np.abs(o["beta"][nz]))

# step by step (host-driven session: the counter is read after every step)
print("host-driven session, k = 10: step, logl rel diff to the oracle, guard fired in this step's score")
o = O.fit_iht(ox, y, None, k=10, dist="poisson", link="log")
ol = np.asarray(o["logl_trace"])
sess = mih.IHTSession(y, x, None, k=10, d=mih.Poisson(), l=mih.LogLink(), step_mode=1)
mih.profile_counters(x, reset=True)
for i in range(120):
    logl, nbt, tol = sess.step()
    c = mih.profile_counters(x, reset=True)
    if 70 <= i <= 100 or i % 10 == 0:
        print(f"  step {i + 1:3d} rel {abs(logl - ol[i]) / abs(ol[i]):.2e} peeled {c['peeled_residuals']} bt {nbt} tol {tol:.3e}")
sess.close()
