"""Which call leaks?  Each variant repeats [new matrix, one kind of call, delete the matrix] 40 times and reports the growth of
the device memory in use (tools/leak_check.py runs them all together)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tests")
import torch, mendeliht_amd as m
from conftest import make_bed
from mendeliht_amd import hash_folds
rng = np.random.default_rng(0)
n, p = 2000, 1500
cols = make_bed(rng, n, p, 0.02)
y = rng.standard_normal(n); Y = rng.standard_normal((2, n)); folds = hash_folds(n, 3)
def used():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
torch.zeros(1, device="cuda")
tests = {
 "matrix only": lambda x: None,
 "fit": lambda x: m.fit_iht(y, x, None, k=5, verbose=False),
 "fit_debias_initbeta": lambda x: m.fit_iht(y, x, None, k=5, verbose=False, debias=True, init_beta=True),
 "cv": lambda x: m.cv_iht(y, x, None, path=[2, 4], q=3, folds=folds, verbose=False),
 "mvfit": lambda x: m.fit_iht(Y, x, None, k=4, verbose=False),
 "mvcv": lambda x: m.cv_iht(Y, x, None, path=[2, 4], q=3, folds=folds, verbose=False),
 "session": lambda x: (lambda s: (s.step(), s.close()))(m.IHTSession(y, x, None, k=3)),
}
for name, fn in tests.items():
    def once():
        x = m.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True); fn(x); del x
    once(); u0 = used()
    for _ in range(40): once()
    print(f"{name}: {used() - u0:+.1f} MiB over 40 rounds", flush=True)
