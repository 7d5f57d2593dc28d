import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tests")
import torch, mendeliht_amd as m
from conftest import make_bed
from mendeliht_amd import hash_folds
rng = np.random.default_rng(0)
n, p = 2000, 1500
cols = make_bed(rng, n, p, 0.02)
y = rng.standard_normal(n)
Y = rng.standard_normal((2, n))
folds = hash_folds(n, 3)
def used():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
torch.zeros(1, device="cuda")
base = None
for rep in range(6):
    for _ in range(20):
        x = m.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)
        m.fit_iht(y, x, None, k=5, verbose=False)
        m.fit_iht(y, x, None, k=5, verbose=False, debias=True, init_beta=True)
        m.cv_iht(y, x, None, path=[2, 4], q=3, folds=folds, verbose=False)
        m.fit_iht(Y, x, None, k=4, verbose=False)
        m.cv_iht(Y, x, None, path=[2, 4], q=3, folds=folds, verbose=False)
        s = m.IHTSession(y, x, None, k=3); s.step(); s.close()
        del x
    u = used()
    base = base if base is not None else u
    print(f"rep {rep}: device memory in use {u:.0f} MiB (delta {u - base:+.0f})", flush=True)
