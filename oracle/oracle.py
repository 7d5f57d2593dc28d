"""ctypes front-end to the CPU oracle (oracle/libiht_oracle.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libiht_oracle.so")

NORMAL, BERNOULLI, POISSON, NEGBIN, GAMMA, INVGAUSS = 0, 1, 2, 3, 4, 5
IDENTITY, LOGIT, LOG, PROBIT, CLOGLOG, CAUCHIT, INVERSE, INVSQUARE, SQRT = range(9)
DIST = {"normal": NORMAL, "bernoulli": BERNOULLI, "poisson": POISSON, "negbin": NEGBIN, "gamma": GAMMA,
        "invgauss": INVGAUSS}
LINK = {"identity": IDENTITY, "logit": LOGIT, "log": LOG, "probit": PROBIT, "cloglog": CLOGLOG, "cauchit": CAUCHIT,
        "inverse": INVERSE, "invsquare": INVSQUARE, "sqrt": SQRT}


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("iht_oracle.c", "iht_oracle_mv.inc", "iht_oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


class _Params(C.Structure):
    _fields_ = [("k", C.c_int64), ("J", C.c_int64), ("dist", C.c_int), ("link", C.c_int),
                ("nb_r", C.c_double), ("tol", C.c_double),
                ("max_iter", C.c_int32), ("min_iter", C.c_int32), ("max_step", C.c_int32),
                ("est_r", C.c_int32),
                ("zkeep", C.c_void_p), ("weight", C.c_void_p), ("group", C.c_void_p),
                ("ks", C.c_void_p), ("nks", C.c_int64), ("init_beta", C.c_int32), ("debias", C.c_int32),
                ("choose", C.c_void_p), ("choose_user", C.c_void_p), ("cv_threads", C.c_int32)]


_CHOOSE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_int64, C.c_int64, C.POINTER(C.c_int64))


def _choose_callback(fn):
    """orc_params.choose from fn(kind, list, excess) -> positions (iht_oracle.h): the draw _choose! makes with the caller's RNG."""
    def cb(_user, kind, lst, n, excess, out):
        try:
            got = np.asarray(fn(int(kind), np.array(lst[:n], dtype=np.int64), int(excess)), dtype=np.int64).ravel()
            want = excess if kind == 0 else n
            if got.size != want:
                return 1
            for t in range(want):
                out[t] = int(got[t])
            return 0
        except Exception:
            return 1
    return _CHOOSE(cb)


class _Result(C.Structure):
    _fields_ = [("logl", C.c_double), ("iter", C.c_int64), ("pve", C.c_double),
                ("nb_r", C.c_double), ("choose_fired", C.c_int32), ("n_trace", C.c_int32),
                ("beta", C.c_void_p), ("c", C.c_void_p), ("logl_trace", C.c_void_p),
                ("tol_trace", C.c_void_p), ("bt_trace", C.c_void_p), ("mu", C.c_void_p), ("eta_cond", C.c_double), ("ib_cond", C.c_double), ("bt_cond", C.c_double), ("db_minstep", C.c_double)]


class _MvResult(C.Structure):
    _fields_ = [("logl", C.c_double), ("iter", C.c_int64), ("n_trace", C.c_int32),
                ("choose_fired", C.c_int32),
                ("B", C.c_void_p), ("C", C.c_void_p), ("Sigma", C.c_void_p), ("pve", C.c_void_p),
                ("logl_trace", C.c_void_p), ("tol_trace", C.c_void_p), ("bt_trace", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_snp_create.restype = C.c_void_p
        _lib.orc_snp_create.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]
        _lib.orc_dense_create.restype = C.c_void_p
        _lib.orc_dense_create.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        _lib.orc_mat_destroy.argtypes = [C.c_void_p]
        _lib.orc_mat_mu_sinv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_getindex.restype = C.c_double
        _lib.orc_getindex.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        _lib.orc_xtv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_xtv_colwise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_xtv_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        _lib.orc_xv_masked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_project_k.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        _lib.orc_project_group_sparse.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int]
        for f in ("orc_linkinv", "orc_mueta"):
            getattr(_lib, f).restype = C.c_double
            getattr(_lib, f).argtypes = [C.c_int, C.c_double]
        _lib.orc_glmvar.restype = C.c_double
        _lib.orc_glmvar.argtypes = [C.c_int, C.c_double, C.c_double]
        _lib.orc_devresid.restype = C.c_double
        _lib.orc_devresid.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
        _lib.orc_loglik_obs.restype = C.c_double
        _lib.orc_loglik_obs.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
        _lib.orc_loglikelihood.restype = C.c_double
        _lib.orc_loglikelihood.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        _lib.orc_deviance.restype = C.c_double
        _lib.orc_deviance.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        _lib.orc_fit_iht.argtypes = [C.c_void_p, C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.POINTER(_Result)]
        _lib.orc_cv_iht.argtypes = [C.c_void_p, C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _lib.orc_fit_mv.argtypes = [C.c_void_p, C.POINTER(_Params), C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int64, C.c_void_p, C.POINTER(_MvResult)]
        _lib.orc_cv_mv.argtypes = [C.c_void_p, C.POINTER(_Params), C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_void_p]
        _lib.orc_set_threads.argtypes = [C.c_int]
        _lib.orc_debias_glm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        _lib.orc_mle_for_r.restype = C.c_double
        _lib.orc_mle_for_r.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_threads(t):
    lib().orc_set_threads(int(t))


class Mat:
    """A design matrix for the oracle: PLINK 2-bit columns or a dense f64 matrix."""

    def __init__(self, handle, n, p, keep):
        self.h, self.n, self.p, self._keep = handle, n, p, keep

    @classmethod
    def from_bed_columns(cls, cols, n, center=True, scale=True, impute=True):
        cols = np.ascontiguousarray(cols, dtype=np.uint8)
        p, stride = cols.shape
        h = lib().orc_snp_create(_p(cols), n, p, stride, int(center), int(scale), int(impute))
        return cls(h, n, p, cols)

    @classmethod
    def from_bed_file(cls, path, n, **kw):
        return cls.from_bed_columns(read_bed(path, n), n, **kw)

    @classmethod
    def from_dense(cls, x):
        x = np.asfortranarray(x, dtype=np.float64)
        n, p = x.shape
        return cls(lib().orc_dense_create(_p(x), n, p), n, p, x)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mat_destroy(self.h)
            self.h = None

    def mu_sinv(self):
        mu, s = np.empty(self.p), np.empty(self.p)
        lib().orc_mat_mu_sinv(self.h, _p(mu), _p(s))
        return mu, s

    def getindex(self, i, j):
        return lib().orc_getindex(self.h, i, j)

    def xtv(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        out = np.empty(self.p)
        lib().orc_xtv(self.h, _p(r), _p(out))
        return out

    def xtv_colwise(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        out = np.empty(self.p)
        lib().orc_xtv_colwise(self.h, _p(r), _p(out))
        return out

    def xtv_multi(self, R):
        R = np.asfortranarray(R, dtype=np.float64)
        m = R.shape[1]
        out = np.empty((self.p, m), order="F")
        lib().orc_xtv_multi(self.h, _p(R), m, _p(out))
        return out

    def xv_masked(self, idx, coef):
        idx = np.ascontiguousarray(idx, dtype=np.uint8)
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        out = np.empty(self.n)
        lib().orc_xv_masked(self.h, _p(idx), _p(coef), _p(out))
        return out


def read_bed(path, n):
    """PLINK .bed (SNP-major) -> (p, ceil(n/4)) uint8 array of column bytes."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw[0] != 0x6C or raw[1] != 0x1B or raw[2] != 0x01:
        raise ValueError("not a SNP-major PLINK .bed file")
    stride = (n + 3) // 4
    body = raw[3:]
    if body.size % stride:
        raise ValueError("bed size does not match n")
    return body.reshape(-1, stride)


def project_k(x, k):
    x = np.array(x, dtype=np.float64)
    rc = lib().orc_project_k(_p(x), x.size, int(k))
    if rc:
        raise ValueError(f"project_k failed rc={rc}")
    return x


def project_group_sparse(y, group, J, k):
    y = np.array(y, dtype=np.float64)
    group = np.ascontiguousarray(group, dtype=np.int64)
    kv = np.ascontiguousarray(np.atleast_1d(k), dtype=np.int64)
    rc = lib().orc_project_group_sparse(_p(y), _p(group), y.size, int(J), _p(kv), int(np.ndim(k) > 0))
    if rc:
        raise ValueError(f"project_group_sparse failed rc={rc}")
    return y


def _params(k, J, dist, link, nb_r, tol, max_iter, min_iter, max_step, est_r, zkeep, weight, group, keep, init_beta=False,
            debias=False, choose=None):
    prm = _Params()
    if choose is not None:
        ccb = _choose_callback(choose)
        prm.choose = C.cast(ccb, C.c_void_p)
        keep.append(ccb)
    ks = None
    if np.ndim(k) > 0:
        ks = np.ascontiguousarray(k, dtype=np.int64)
        prm.k = 0
    else:
        prm.k = int(k)
    prm.J = int(J)
    prm.dist = DIST[dist] if isinstance(dist, str) else int(dist)
    prm.link = LINK[link] if isinstance(link, str) else int(link)
    prm.nb_r, prm.tol = float(nb_r), float(tol)
    prm.max_iter, prm.min_iter, prm.max_step = int(max_iter), int(min_iter), int(max_step)
    prm.est_r = {None: 0, "none": 0, "mm": 1, "newton": 2}[est_r.lower() if isinstance(est_r, str) else est_r]
    zk = None if zkeep is None else np.ascontiguousarray(zkeep, dtype=np.uint8)
    w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
    g = None if group is None else np.ascontiguousarray(group, dtype=np.int64)
    prm.zkeep, prm.weight, prm.group, prm.ks = _p(zk), _p(w), _p(g), _p(ks)
    prm.nks = 0 if ks is None else ks.size
    prm.init_beta = int(bool(init_beta))
    prm.debias = int(bool(debias))
    keep.extend([zk, w, g, ks])
    return prm


def _z(z, n):
    if z is None:
        z = np.ones((n, 1))
    z = np.asfortranarray(np.asarray(z, dtype=np.float64).reshape(n, -1))
    return z


def fit_iht(x, y, z=None, k=10, J=1, dist="normal", link="identity", nb_r=1.0, tol=1e-4,
            max_iter=200, min_iter=5, max_step=3, est_r=None, zkeep=None, weight=None,
            group=None, train=None, init_beta=False, debias=False, choose=None):
    keep = []
    prm = _params(k, J, dist, link, nb_r, tol, max_iter, min_iter, max_step, est_r, zkeep, weight, group, keep, init_beta,
                  debias, choose)
    y = np.ascontiguousarray(y, dtype=np.float64)
    z = _z(z, x.n)
    q = z.shape[1]
    tr = None if train is None else np.ascontiguousarray(train, dtype=np.uint8)
    beta, c, mu = np.zeros(x.p), np.zeros(q), np.zeros(x.n)
    lt, tt, bt = np.zeros(max_iter + 1), np.zeros(max_iter + 1), np.zeros(max_iter + 1, dtype=np.int32)
    res = _Result()
    res.beta, res.c, res.mu = _p(beta), _p(c), _p(mu)
    res.logl_trace, res.tol_trace, res.bt_trace = _p(lt), _p(tt), _p(bt)
    rc = lib().orc_fit_iht(x.h, C.byref(prm), _p(y), _p(z), q, _p(tr), C.byref(res))
    if rc:
        e = RuntimeError(f"orc_fit_iht rc={rc}")
        e.db_minstep = res.db_minstep                # (diagnostic: was it debias!'s refit, crawling by halved steps?  iht_oracle.h)
        raise e
    nt = res.n_trace
    return dict(logl=res.logl, iter=res.iter, pve=res.pve, beta=beta, c=c, mu=mu, nb_r=res.nb_r,
                choose_fired=bool(res.choose_fired), logl_trace=lt[:nt].copy(), tol_trace=tt[:nt].copy(),
                bt_trace=bt[:nt].copy(), eta_cond=res.eta_cond, ib_cond=res.ib_cond, bt_cond=res.bt_cond, db_minstep=res.db_minstep)


def cv_iht(x, y, z=None, path=range(1, 21), q=5, folds=None, dist="normal", link="identity", nb_r=1.0,
           tol=1e-4, max_iter=100, min_iter=5, max_step=3, est_r=None, zkeep=None, weight=None,
           group=None, J=1, init_beta=False, debias=False, cv_threads=0):
    """cv_threads: Threads.nthreads() of the reference run (matters with est_r only: iht_oracle.h); 0 = 1 = one chain, as in
    mih_fit_params and both bindings."""
    keep = []
    prm = _params(1, J, dist, link, nb_r, tol, max_iter, min_iter, max_step, est_r, zkeep, weight, group, keep, init_beta,
                  debias)
    prm.cv_threads = int(cv_threads)
    y = np.ascontiguousarray(y, dtype=np.float64)
    z = _z(z, x.n)
    folds = np.ascontiguousarray(folds, dtype=np.int32)
    path = np.ascontiguousarray(list(path), dtype=np.int64)
    raw, mse = np.zeros(q * path.size), np.zeros(path.size)
    rc = lib().orc_cv_iht(x.h, C.byref(prm), _p(y), _p(z), z.shape[1], _p(folds), q, _p(path), path.size,
                          _p(raw), _p(mse))
    if rc:
        raise RuntimeError(f"orc_cv_iht rc={rc}")
    return mse, raw.reshape(q, path.size)


def fit_mv(x, Y, Z=None, k=10, tol=1e-4, max_iter=200, min_iter=5, max_step=3, zkeep=None, train=None,
           init_beta=False, choose=None):
    """Y is r x n (traits x samples), Z is q x n; returns B (r x p), C (r x q)."""
    keep = []
    prm = _params(k, 1, "normal", "identity", 1.0, tol, max_iter, min_iter, max_step, None, zkeep, None, None, keep,
                  init_beta, choose=choose)
    Y = np.asfortranarray(Y, dtype=np.float64)
    r, n = Y.shape
    if Z is None:
        Z = np.ones((1, n))
    Z = np.asfortranarray(Z, dtype=np.float64)
    q = Z.shape[0]
    tr = None if train is None else np.ascontiguousarray(train, dtype=np.uint8)
    B, Cm = np.zeros((r, x.p), order="F"), np.zeros((r, q), order="F")
    S, pve = np.zeros((r, r), order="F"), np.zeros(r)
    lt, tt, bt = np.zeros(max_iter + 1), np.zeros(max_iter + 1), np.zeros(max_iter + 1, dtype=np.int32)
    res = _MvResult()
    res.B, res.C, res.Sigma, res.pve = _p(B), _p(Cm), _p(S), _p(pve)
    res.logl_trace, res.tol_trace, res.bt_trace = _p(lt), _p(tt), _p(bt)
    rc = lib().orc_fit_mv(x.h, C.byref(prm), _p(Y), r, _p(Z), q, _p(tr), C.byref(res))
    if rc:
        raise RuntimeError(f"orc_fit_mv rc={rc}")
    nt = res.n_trace
    return dict(logl=res.logl, iter=res.iter, B=B, C=Cm, Sigma=S, pve=pve, choose_fired=bool(res.choose_fired),
                logl_trace=lt[:nt].copy(), tol_trace=tt[:nt].copy(), bt_trace=bt[:nt].copy())


def cv_mv(x, Y, Z=None, path=range(1, 21), q=5, folds=None, tol=1e-4, max_iter=100, min_iter=5,
          max_step=3, zkeep=None, init_beta=False):
    keep = []
    prm = _params(1, 1, "normal", "identity", 1.0, tol, max_iter, min_iter, max_step, None, zkeep, None, None, keep,
                  init_beta)
    Y = np.asfortranarray(Y, dtype=np.float64)
    r, n = Y.shape
    if Z is None:
        Z = np.ones((1, n))
    Z = np.asfortranarray(Z, dtype=np.float64)
    folds = np.ascontiguousarray(folds, dtype=np.int32)
    path = np.ascontiguousarray(list(path), dtype=np.int64)
    raw, mse = np.zeros(q * path.size), np.zeros(path.size)
    rc = lib().orc_cv_mv(x.h, C.byref(prm), _p(Y), r, _p(Z), Z.shape[0], _p(folds), q, _p(path), path.size,
                         _p(raw), _p(mse))
    if rc:
        raise RuntimeError(f"orc_cv_mv rc={rc}")
    return mse, raw.reshape(q, path.size)


_DIST = {"normal": 0, "bernoulli": 1, "poisson": 2, "negbin": 3, "gamma": 4, "invgauss": 5}
_LINK = {"identity": 0, "logit": 1, "log": 2, "probit": 3, "cloglog": 4, "cauchit": 5, "inverse": 6, "invsquare": 7, "sqrt": 8}


def debias_glm(x, support_mask, y, dist, link, nb_r=1.0):
    """debias! (utilities.jl:1014-1020) on its own: GLM refit of y on the columns of the mask; returns the length-p coefficient vector."""
    m = np.ascontiguousarray(support_mask, dtype=np.uint8)
    y = np.ascontiguousarray(y, dtype=np.float64)
    b = np.zeros(x.p)
    rc = lib().orc_debias_glm(x.h, _p(m), _p(y), _DIST[dist], _LINK[link], float(nb_r), _p(b))
    if rc:
        raise RuntimeError(f"orc_debias_glm rc={rc}")
    return b


def mle_for_r(y, mu, r0=1.0, method="newton"):
    """mle_for_r (utilities.jl:141-247) on its own: the NegBin r given y and mu."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    return lib().orc_mle_for_r(_p(y), _p(mu), None, y.size, float(r0), {"mm": 1, "newton": 2}[method])


def standardize_columns(z):
    """standardize! (utilities.jl:494-530): (z - mean) / sample-sd per column."""
    z = np.array(z, dtype=np.float64)
    mu = z.mean(axis=0)
    sd = np.sqrt(((z - mu) ** 2).sum(axis=0) / (z.shape[0] - 1))
    return (z - mu) / sd
