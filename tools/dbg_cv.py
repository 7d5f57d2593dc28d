import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mendeliht_amd as m
from oracle import oracle as O
from conftest import hash_folds
n = 1000
bed = m.read_bed(os.path.join(ROOT, "tests/fixtures/normal.bed"), n)
x = m.SnpLinAlg(bed, n, center=True, scale=True, impute=True); ox = O.Mat.from_bed_columns(bed, n)
rng = np.random.default_rng(20)
p = ox.p; b = np.zeros(p); supp = rng.choice(p, 6, replace=False); b[supp] = rng.standard_normal(6) * 0.7
mask = np.zeros(p, np.uint8); mask[supp] = 1
eta = ox.xv_masked(mask, b)
y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = hash_folds(n, 3)
train = (folds != 1).astype(np.uint8)
r = m.fit_iht(y, x, None, k=4, d=m.Bernoulli(), l=m.LogitLink(), verbose=False, train=train, max_iter=100)
o = O.fit_iht(ox, y, None, k=4, dist="bernoulli", link="logit", train=train, max_iter=100)
print("iters", r.iter, o["iter"])
for i in range(max(len(r.trace["logl"]), len(o["logl_trace"]))):
    a = (r.trace["logl"][i], r.trace["backtracks"][i], r.trace["tol"][i]) if i < len(r.trace["logl"]) else None
    c = (o["logl_trace"][i], o["bt_trace"][i], o["tol_trace"][i]) if i < len(o["logl_trace"]) else None
    print(i + 1, a, c)
print(np.flatnonzero(r.beta), np.flatnonzero(o["beta"]))
