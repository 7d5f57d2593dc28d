"""First-contact GPU check: parity of every kernel family vs the oracle + variant timing."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mendeliht_amd as m
from oracle import oracle as O
from conftest import make_bed, hash_folds

def rel(a, b): return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))

print("devices", m.device_count())
rng = np.random.default_rng(1)
n = 1000
bed = m.read_bed(os.path.join(ROOT, "tests/fixtures/normal.bed"), n)
x = m.SnpLinAlg(bed, n, center=True, scale=True, impute=True)
ox = O.Mat.from_bed_columns(bed, n)
mu, s = x.mu_sigma(); omu, os_ = ox.mu_sinv()
print("mu/sinv", rel(mu, omu), rel(s, os_))
r = rng.standard_normal(n)
for v in range(14):
    m.lib().mih_set_xtv_variant(v)
    print("xtv variant", v, rel(x.xtv(r), ox.xtv(r)))
m.lib().mih_set_xtv_variant(-1)
# ragged n + missing
for (nn, pp, mr) in [(1003, 257, 0.02), (77, 33, 0.1), (5000, 100, 0.0), (2049, 64, 0.05)]:
    cols = make_bed(rng, nn, pp, mr)
    for (c_, s_, i_) in [(1, 1, 1), (1, 1, 0), (0, 0, 1), (1, 0, 1)]:
        xs = m.SnpLinAlg(cols, nn, center=c_, scale=s_, impute=i_)
        oxs = O.Mat.from_bed_columns(cols, nn, center=c_, scale=s_, impute=i_)
        rr = rng.standard_normal(nn)
        e1 = rel(xs.xtv(rr), oxs.xtv(rr))
        idx = np.sort(rng.choice(pp, size=min(7, pp), replace=False)); val = rng.standard_normal(idx.size)
        mask = np.zeros(pp, np.uint8); mask[idx] = 1; coef = np.zeros(pp); coef[idx] = val
        e2 = rel(xs.xv_sparse(idx, val), oxs.xv_masked(mask, coef))
        e3 = np.array_equal(xs.export_bed(), cols) if mr == 0 or True else None
        print("ragged", nn, pp, mr, (c_, s_, i_), e1, e2, e3)
# topk
v = rng.standard_normal(100000)
pk = m.project_k(v, 100); ok = O.project_k(v, 100)
print("topk equal", np.array_equal(pk, ok), np.count_nonzero(pk))
# fit G1
y = np.loadtxt(os.path.join(ROOT, "tests/fixtures/normal_y_fam6.txt"))
z = np.loadtxt(os.path.join(ROOT, "tests/fixtures/covariates.txt"), delimiter=","); z[:, 1:] = m.standardize(z[:, 1:])
t = time.time(); res = m.fit_iht(y, x, z, k=7, verbose=True); print("fit time", time.time() - t, res.time)
o = O.fit_iht(ox, y, z, k=7)
print("G1 logl", res.logl, o["logl"], "iter", res.iter, o["iter"], "beta rel", rel(res.beta, o["beta"]), "support", np.array_equal(np.flatnonzero(res.beta), np.flatnonzero(o["beta"])), "c", res.c, "pve", res.σg, o["pve"])
# logistic / poisson
b = np.zeros(10000); supp = rng.choice(10000, 8, replace=False); b[supp] = rng.standard_normal(8) * 0.5
mask = np.zeros(10000, np.uint8); mask[supp] = 1
eta = ox.xv_masked(mask, b)
yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
yp = rng.poisson(np.exp(0.3 * eta)).astype(float)
for (name, yy, dd, ll, od, ol) in [("logit", yb, m.Bernoulli(), m.LogitLink(), "bernoulli", "logit"), ("poisson", yp, m.Poisson(), m.LogLink(), "poisson", "log")]:
    res = m.fit_iht(yy, x, None, k=8, d=dd, l=ll, verbose=False)
    o = O.fit_iht(ox, yy, None, k=8, dist=od, link=ol)
    print(name, "logl", res.logl, o["logl"], "iter", res.iter, o["iter"], "beta rel", rel(res.beta, o["beta"]), "support", np.array_equal(np.flatnonzero(res.beta), np.flatnonzero(o["beta"])), "bt", res.trace["backtracks"].sum(), o["bt_trace"].sum())
# cv
folds = hash_folds(n, 3)
t = time.time(); mse, raw = m.cv_iht(y, x, z, path=range(0, 6), q=3, folds=folds, verbose=False, return_raw=True); print("cv time", time.time() - t)
omse, oraw = O.cv_iht(ox, y, z, path=range(0, 6), q=3, folds=folds)
print("cv rel", rel(mse, omse), rel(raw, oraw))
# timing of variants on a mid-size synthetic matrix
xs = m.SnpLinAlg.synthetic(100000, 65536, seed=3)
bytes_ = xs.algorithmic_bytes()
for v in range(14):
    ms, cs = xs.bench_xtv(v, iters=5, warmup=1)
    print(f"variant {v}: {ms:.3f} ms  {bytes_/ms/1e9:.1f} GB/s checksum {cs:.6e}")
