"""Bad arguments at the C ABI: every call below must come back with a non-zero status (and a message), never crash, and leave the
library usable.  Each call is announced before it runs, so a crash names its culprit; run as a child process by
tests/test_gpu_parity.py::test_c_abi_refuses_bad_arguments_without_crashing."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mendeliht_amd as m
from mendeliht_amd import api
from conftest import make_bed

L = api.lib()
vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
NULL = None
bad = []


def refuse(name, rc):
    print(f"{name}: rc={rc}", flush=True)
    if rc == 0:
        bad.append(name)


def call(name, fn, *args):
    print(f"-> {name}", flush=True)
    refuse(name, fn(*args))


rng = np.random.default_rng(0)
n, p = 300, 64
cols = make_bed(rng, n, p)
x = m.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)
h = x._h
out_h = vp()
dbl = lambda a: a.ctypes.data_as(vp)
y = rng.standard_normal(n); z = np.ones((n, 1)); r = rng.standard_normal(n); outp = np.zeros(p); outn = np.zeros(n)
stride = cols.shape[1]

# ---- matrices
call("snp_create NULL columns", L.mih_snp_create, NULL, n, p, stride, 1, 1, 1, 64, 0, C.byref(out_h))
call("snp_create n = 0", L.mih_snp_create, dbl(cols), 0, p, stride, 1, 1, 1, 64, 0, C.byref(out_h))
call("snp_create p = 0", L.mih_snp_create, dbl(cols), n, 0, stride, 1, 1, 1, 64, 0, C.byref(out_h))
call("snp_create n < 0", L.mih_snp_create, dbl(cols), -5, p, stride, 1, 1, 1, 64, 0, C.byref(out_h))
call("snp_create stride too small", L.mih_snp_create, dbl(cols), n, p, stride - 1, 1, 1, 1, 64, 0, C.byref(out_h))
call("snp_create dtype 16", L.mih_snp_create, dbl(cols), n, p, stride, 1, 1, 1, 16, 0, C.byref(out_h))
call("snp_create device 99", L.mih_snp_create, dbl(cols), n, p, stride, 1, 1, 1, 64, 99, C.byref(out_h))
call("snp_create NULL out", L.mih_snp_create, dbl(cols), n, p, stride, 1, 1, 1, 64, 0, NULL)
call("dense_create NULL", L.mih_dense_create, NULL, n, p, 0, C.byref(out_h))
call("dense_create n = 0", L.mih_dense_create, dbl(np.zeros((4, 4))), 0, 4, 0, C.byref(out_h))
call("synthetic p = 0", L.mih_snp_create_synthetic, n, 0, 1, 0.0, 1, 1, 1, 0, C.byref(out_h))
call("synthetic missing rate 2", L.mih_snp_create_synthetic, n, p, 1, C.c_double(2.0), 1, 1, 1, 0, C.byref(out_h))
call("mat_dims NULL handle", L.mih_mat_dims, NULL, C.byref(i64()), C.byref(i64()))
call("mu_sigma NULL handle", L.mih_snp_mu_sigma, NULL, dbl(outp), dbl(outp))
# ---- linear algebra
call("xtv NULL r", L.mih_xtv, h, NULL, dbl(outp))
call("xtv NULL out", L.mih_xtv, h, dbl(r), NULL)
call("xtv NULL handle", L.mih_xtv, NULL, dbl(r), dbl(outp))
call("xtv_batched m = 0", L.mih_xtv_batched, h, dbl(r), 0, dbl(outp))
call("xtv_batched m = -1", L.mih_xtv_batched, h, dbl(r), -1, dbl(outp))
call("xtv_batched_fmt digits 999", L.mih_xtv_batched_fmt, h, dbl(r), 1, 999, dbl(outp))
idx = np.array([0, 5, p], dtype=np.int64); val = np.ones(3)
call("xv_sparse index = p", L.mih_xv_sparse, h, dbl(idx), dbl(val), 3, dbl(outn))
idx = np.array([-1, 5], dtype=np.int64)
call("xv_sparse index < 0", L.mih_xv_sparse, h, dbl(idx), dbl(val), 2, dbl(outn))
call("xv_sparse nnz < 0", L.mih_xv_sparse, h, dbl(idx), dbl(val), -1, dbl(outn))
call("xv_sparse NULL idx, nnz 2", L.mih_xv_sparse, h, NULL, dbl(val), 2, dbl(outn))
# ---- projections
v = rng.standard_normal(50); kept = i64()
call("project_topk NULL", L.mih_project_topk, NULL, 50, 3, C.byref(kept))
call("project_topk k = 0", L.mih_project_topk, dbl(v), 50, 0, C.byref(kept))
call("project_topk k < 0", L.mih_project_topk, dbl(v), 50, -2, C.byref(kept))
call("project_topk k > len", L.mih_project_topk, dbl(v), 50, 51, C.byref(kept))
call("project_topk len = 0", L.mih_project_topk, dbl(v), 0, 1, C.byref(kept))
g = np.repeat(np.arange(1, 6), 10).astype(np.int64); one = np.array([2], dtype=np.int64)
g0 = g.copy(); g0[3] = 0
call("group_sparse label 0", L.mih_project_group_sparse, dbl(v), dbl(g0), 50, 2, dbl(one), 0)
gm = g.copy(); gm[7] = -4
call("group_sparse label < 0", L.mih_project_group_sparse, dbl(v), dbl(gm), 50, 2, dbl(one), 0)
call("group_sparse J < 0", L.mih_project_group_sparse, dbl(v), dbl(g), 50, -1, dbl(one), 0)
call("group_sparse k < 0", L.mih_project_group_sparse, dbl(v), dbl(g), 50, 2, dbl(np.array([-1], dtype=np.int64)), 0)
call("group_sparse NULL group", L.mih_project_group_sparse, dbl(v), NULL, 50, 2, dbl(one), 0)
call("group_sparse NULL k", L.mih_project_group_sparse, dbl(v), dbl(g), 50, 2, NULL, 0)
# ---- fits
def params(**kw):
    prm = api._FitParams()
    prm.k, prm.J, prm.dist, prm.link, prm.nb_r, prm.tol = 3, 1, 0, 0, 1.0, 1e-4
    prm.max_iter, prm.min_iter, prm.max_step, prm.est_r = 50, 5, 3, 0
    for key, val_ in kw.items():
        setattr(prm, key, val_)
    return prm
beta = np.zeros(p); c = np.zeros(1); lt = np.zeros(60); tt = np.zeros(60); bt = np.zeros(60, dtype=np.int32)
def result():
    res = api._FitResult()
    res.beta, res.c, res.logl_trace, res.tol_trace, res.bt_trace = dbl(beta), dbl(c), dbl(lt), dbl(tt), dbl(bt)
    return res
fit = lambda name, prm, yy=dbl(y), zz=dbl(z), q=1, hh=h, res=None: call(name, L.mih_fit_iht, hh, C.byref(prm) if prm is not None else NULL, yy, zz, q, NULL, C.byref(res or result()))
fit("fit NULL params", None)
fit("fit NULL y", params(), yy=NULL)
fit("fit NULL z", params(), zz=NULL)
fit("fit q = 0", params(), q=0)
fit("fit NULL handle", params(), hh=NULL)
fit("fit k < 0", params(k=-1))
fit("fit k > p + q", params(k=p + 5))
fit("fit J < 0", params(J=-1))
fit("fit dist 77", params(dist=77))
fit("fit link 77", params(link=77))
fit("fit tol 0", params(tol=0.0))
fit("fit max_iter < 0", params(max_iter=-3))
fit("fit max_step < 0", params(max_step=-1))
fit("fit est_r 9", params(dist=3, link=2, est_r=9))
fit("fit est_r on Normal", params(est_r=1))
fit("fit xtv_digits 7", params(xtv_digits=7))
gbad = np.zeros(p, dtype=np.int64)
fit("fit group labels 0", params(group=dbl(gbad)))
ksv = np.array([1, -1, 2], dtype=np.int64); gg = (np.arange(p) % 3 + 1).astype(np.int64)
fit("fit ks with a negative entry", params(group=dbl(gg), ks=dbl(ksv), nks=3, J=2))
fit("fit ks shorter than the groups", params(group=dbl(gg), ks=dbl(ksv[:1]), nks=1, J=2))
fit("fit ks without groups", params(ks=dbl(np.array([1, 2], dtype=np.int64)), nks=2))
wneg = -np.ones(p)
yb = np.full(n, 0.5)
print("-> fit Bernoulli with y = 0.5 (accepted: checky is the binding's job, fit.jl:91 -- GLM.checky in the glue, _checky in the mirror)", flush=True)
assert L.mih_fit_iht(h, C.byref(params(dist=1, link=1)), dbl(yb), dbl(z), 1, NULL, C.byref(result())) == 0
ynan = y.copy(); ynan[3] = np.nan
fit("fit y with NaN", params(), yy=dbl(ynan))
call("fit NULL result", L.mih_fit_iht, h, C.byref(params()), dbl(y), dbl(z), 1, NULL, NULL)
# ---- cross-validation, paths
folds = (np.arange(n) % 3 + 1).astype(np.int32); path = np.array([1, 2, 3], dtype=np.int64); raw = np.zeros(9)
cv = lambda name, f=dbl(folds), nf=3, pa=dbl(path), np_=3, rank=0, world=1, prm=None: call(name, L.mih_cv_iht, h, C.byref(prm or params()), dbl(y), dbl(z), 1, f, nf, pa, np_, rank, world, dbl(raw))
cv("cv nfolds = 0", nf=0)
cv("cv npath = 0", np_=0)
cv("cv NULL folds", f=NULL)
cv("cv NULL path", pa=NULL)
fb = folds.copy(); fb[5] = 9
cv("cv fold label out of range", f=dbl(fb))
fb = folds.copy(); fb[5] = 0
cv("cv fold label 0", f=dbl(fb))
print("-> cv path entry 0 (accepted: k = 0 is a legal model size, fit.jl:87 asks for k >= 0)", flush=True)
assert L.mih_cv_iht(h, C.byref(params()), dbl(y), dbl(z), 1, dbl(folds), 3, dbl(np.array([0, 1, 2], dtype=np.int64)), 3, 0, 1, dbl(raw)) == 0
cv("cv path entry < 0", pa=dbl(np.array([-1, 1, 2], dtype=np.int64)))
cv("cv path entry > p", pa=dbl(np.array([1, 2, p + 9], dtype=np.int64)))
cv("cv rank = world", rank=2, world=2)
cv("cv world = 0", world=0)
cv("cv rank < 0", rank=-1, world=2)
ro = np.zeros(9, dtype=np.int32)
call("cv_assignment world = 0", L.mih_cv_assignment, dbl(path), 3, 3, 0, dbl(ro))
call("cv_assignment NULL", L.mih_cv_assignment, NULL, 3, 3, 2, dbl(ro))
ll = np.zeros(3)
call("path npath = 0", L.mih_fit_iht_path, h, C.byref(params()), dbl(y), dbl(z), 1, dbl(path), 0, 0, 1, dbl(ll), NULL, NULL, NULL)
call("path rank = world", L.mih_fit_iht_path, h, C.byref(params()), dbl(y), dbl(z), 1, dbl(path), 3, 3, 3, dbl(ll), NULL, NULL, NULL)
call("path NULL logl", L.mih_fit_iht_path, h, C.byref(params()), dbl(y), dbl(z), 1, dbl(path), 3, 0, 1, NULL, NULL, NULL, NULL)
# ---- multivariate
Y = np.asfortranarray(rng.standard_normal((2, n))); Z = np.ones((1, n))
mres = api._MvResult()
B = np.zeros((2, p), order="F"); Cm = np.zeros((2, 1), order="F"); S = np.zeros((2, 2), order="F"); pve = np.zeros(2)
mres.B, mres.C, mres.Sigma, mres.pve, mres.logl_trace, mres.tol_trace, mres.bt_trace = dbl(B), dbl(Cm), dbl(S), dbl(pve), dbl(lt), dbl(tt), dbl(bt)
call("fit_mv r = 0", L.mih_fit_mv, h, C.byref(params()), dbl(Y), 0, dbl(Z), 1, NULL, C.byref(mres))
call("fit_mv r = 1000", L.mih_fit_mv, h, C.byref(params()), dbl(Y), 1000, dbl(Z), 1, NULL, C.byref(mres))
call("fit_mv NULL Y", L.mih_fit_mv, h, C.byref(params()), NULL, 2, dbl(Z), 1, NULL, C.byref(mres))
call("fit_mv k > r (p + q)", L.mih_fit_mv, h, C.byref(params(k=2 * (p + 1) + 1)), dbl(Y), 2, dbl(Z), 1, NULL, C.byref(mres))
# ---- sessions, hooks
sess = vp()
call("session_create NULL handle", L.mih_session_create, NULL, C.byref(params()), dbl(y), dbl(z), 1, NULL, C.byref(sess))
call("session_create NULL out", L.mih_session_create, h, C.byref(params()), dbl(y), dbl(z), 1, NULL, NULL)
call("session_step NULL", L.mih_session_step, NULL, C.byref(C.c_double()), C.byref(i32()), C.byref(C.c_double()))
call("session_model NULL", L.mih_session_model, NULL, dbl(beta), dbl(c))
call("profile_enable NULL", L.mih_profile_enable, NULL, 1)
call("abi_sizes NULL", L.mih_abi_sizes, NULL, 4)
call("bench_xtv m = 0", L.mih_bench_xtv, h, 0, 0, 1, 0, 1, C.byref(C.c_float()), C.byref(C.c_double()))
print("-> mat_destroy(NULL), session_destroy(NULL), last_error(NULL, 0): must not crash", flush=True)
L.mih_mat_destroy(NULL); L.mih_session_destroy(NULL); L.mih_last_error(NULL, 0)
# ---- the library still works
ok = m.fit_iht(y, x, None, k=3, verbose=False)
assert np.count_nonzero(ok.beta) == 3
print("ACCEPTED:", bad, flush=True)
print("probe finished", flush=True)
sys.exit(1 if bad else 0)
