"""Host-side mirror of MendelIHT.jl's hot-path API over the C ABI (ctypes).

Names, keyword arguments, defaults and error behaviour follow the reference:
  fit_iht   src/fit.jl:60-118          cv_iht          src/cross_validation.jl:60-131
  iht       src/wrapper.jl:52-120      cross_validate  src/wrapper.jl:301-349
  project_k! / project_group_sparse!   src/utilities.jl:553-559, 613-679
  SnpLinAlg (SnpArrays.jl)             constructed as in src/wrapper.jl:68-69
All numerics run in libmendeliht_hip.so on the GPU; this file only marshals.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = "libmendeliht_hip.so"
# the measurement build of the same sources (-DMIH_PROBES: launch-shape sweeps, round-1 kernel families, timing probes, the
# MENDELIHT_* A/B switches).  tools/ and the "this switch changes nothing" tests select it with MENDELIHT_HIP_PROBES=1 in the
# environment (read once, at the first lib() call); the product never does.
_PROBES_LIBNAME = "libmendeliht_hip_probes.so"


class MendelIHTError(RuntimeError):
    """Base class for errors reported by the HIP library."""


class DimensionMismatch(MendelIHTError, ValueError):
    pass


class ArgumentError(MendelIHTError, ValueError):
    pass


_STATUS = {1: DimensionMismatch, 2: ArgumentError, 3: ArgumentError, 4: MendelIHTError, 5: MendelIHTError,
           6: MendelIHTError, 7: MemoryError, 8: MendelIHTError}


def library_path():
    return os.path.join(_HERE, _LIBNAME)


def probes_library_path():
    return os.path.join(_HERE, _PROBES_LIBNAME)


def using_probes():
    e = os.environ.get("MENDELIHT_HIP_PROBES", "")
    return e not in ("", "0")


class _FitParams(C.Structure):
    _fields_ = [("k", C.c_int64), ("J", C.c_int64), ("dist", C.c_int32), ("link", C.c_int32),
                ("nb_r", C.c_double), ("tol", C.c_double),
                ("max_iter", C.c_int32), ("min_iter", C.c_int32), ("max_step", C.c_int32), ("est_r", C.c_int32),
                ("zkeep", C.c_void_p), ("weight", C.c_void_p), ("group", C.c_void_p), ("ks", C.c_void_p),
                ("nks", C.c_int64), ("progress", C.c_void_p), ("progress_user", C.c_void_p),
                ("init_beta", C.c_int32), ("comm", C.c_void_p), ("debias", C.c_int32), ("xtv_digits", C.c_int32),
                ("choose", C.c_void_p), ("choose_user", C.c_void_p), ("cv_threads", C.c_int32), ("step_mode", C.c_int32)]


class _Comm(C.Structure):
    """mih_comm (include/mendeliht_hip.h): the exchange callbacks of a column-sharded fit."""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("col_offset", C.c_int64), ("p_global", C.c_int64),
                ("allreduce", C.c_void_p), ("allgather", C.c_void_p), ("user", C.c_void_p)]


_CHOOSE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_int64, C.c_int64, C.POINTER(C.c_int64))
CHOOSE_SAMPLE, CHOOSE_SHUFFLE_B, CHOOSE_SHUFFLE_C = 0, 1, 2      # mih_fit_params::choose kinds (include/mendeliht_hip.h)


def _choose_callback(fn):
    """mih_fit_params::choose from a Python callable fn(kind, list, excess) -> positions: the caller's stand-in for the
    reference's RNG draw in _choose! (`sample(non_zero_idx, excess, replace=false)`, src/utilities.jl:453; `shuffle!`,
    src/multivariate.jl:336-337).  kind CHOOSE_SAMPLE: return `excess` distinct entries of list; CHOOSE_SHUFFLE_*: return
    the whole list in shuffled order."""
    def cb(_user, kind, lst, n, excess, out):
        try:
            got = np.asarray(fn(int(kind), np.array(lst[:n], dtype=np.int64), int(excess)), dtype=np.int64).ravel()
            want = excess if kind == CHOOSE_SAMPLE else n
            if got.size != want:
                return 1
            for t in range(want):
                out[t] = int(got[t])
            return 0
        except Exception:       # an exception must not unwind through the C frames
            return 1
    return _CHOOSE(cb)


_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)
_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class _FitResult(C.Structure):
    _fields_ = [("time", C.c_double), ("logl", C.c_double), ("iter", C.c_int64), ("pve", C.c_double),
                ("nb_r", C.c_double), ("choose_fired", C.c_int32), ("n_trace", C.c_int32),
                ("beta", C.c_void_p), ("c", C.c_void_p), ("logl_trace", C.c_void_p), ("tol_trace", C.c_void_p),
                ("bt_trace", C.c_void_p), ("mu", C.c_void_p)]


class _MvResult(C.Structure):
    _fields_ = [("time", C.c_double), ("logl", C.c_double), ("iter", C.c_int64),
                ("choose_fired", C.c_int32), ("n_trace", C.c_int32),
                ("B", C.c_void_p), ("C", C.c_void_p), ("Sigma", C.c_void_p), ("pve", C.c_void_p),
                ("logl_trace", C.c_void_p), ("tol_trace", C.c_void_p), ("bt_trace", C.c_void_p)]


class _PassRecord(C.Structure):
    """mih_pass_record: one launch of the dominant X'r kernel as the measurement hook of a matrix recorded it."""
    _fields_ = [("start_ms", C.c_double), ("ms", C.c_double), ("residuals", C.c_int32), ("operands", C.c_int32),
                ("stream_tag", C.c_int32), ("reserved", C.c_int32), ("kernel", C.c_char * 48)]


PROFILE_COUNTERS = ("lanes", "max_in_flight", "handovers", "shared_init", "rounds", "fits", "scores", "max_lane_slots", "init_scores",
                    "resident_steps", "resident_attempts", "resident_handbacks", "resident_direct", "resident_redos", "skipped_last_scores", "residuals_43bit",
                    "peeled_residuals")

_PROGRESS = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double)

_lib = None


def lib():
    """Load the HIP library; fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = probes_library_path() if using_probes() else library_path()
    path = os.environ.get("MENDELIHT_HIP_LIB", path)          # an explicit build of the library (the Julia glue reads the same variable)
    if not os.path.exists(path):
        raise MendelIHTError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    L = C.CDLL(path)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    sig = {
        "mih_device_count": [C.POINTER(C.c_int)],
        "mih_last_error": [C.c_char_p, C.c_size_t],
        "mih_version": [C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "mih_snp_create": [vp, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
        "mih_snp_create_synthetic": [i64, i64, C.c_uint64, dbl, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
        "mih_snp_create_synthetic_shard": [i64, i64, i64, C.c_uint64, dbl, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
        "mih_dense_create": [vp, i64, i64, C.c_int, C.POINTER(vp)],
        "mih_dense_create_synthetic": [i64, i64, C.c_uint64, C.c_int, C.POINTER(vp)],
        "mih_dense_create_f32": [vp, i64, i64, C.c_int, C.POINTER(vp)],
        "mih_mat_destroy": [vp],
        "mih_mat_dims": [vp, C.POINTER(i64), C.POINTER(i64)],
        "mih_mat_reserve": [vp, i64],
        "mih_snp_mu_sigma": [vp, vp, vp],
        "mih_snp_export_bed": [vp, vp],
        "mih_snp_naive_impute": [vp, vp],
        "mih_xtv": [vp, vp, vp],
        "mih_xtv_batched": [vp, vp, C.c_int, vp],
        "mih_xtv_batched_fmt": [vp, vp, C.c_int, C.c_int, vp],
        "mih_xv_sparse": [vp, vp, vp, i64, vp],
        "mih_project_topk": [vp, i64, i64, C.POINTER(i64)],
        "mih_project_group_sparse": [vp, vp, i64, i64, vp, C.c_int],
        "mih_fit_iht": [vp, C.POINTER(_FitParams), vp, vp, i64, vp, C.POINTER(_FitResult)],
        "mih_cv_iht": [vp, C.POINTER(_FitParams), vp, vp, i64, vp, i32, vp, i64, i32, i32, vp],
        "mih_cv_meanloss": [vp, vp, i64, i32, i64, vp],
        "mih_cv_assignment": [vp, i64, i32, i32, vp],
        "mih_fit_iht_path": [vp, C.POINTER(_FitParams), vp, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp],
        "mih_cv_iht_multi": [vp, i32, C.POINTER(_FitParams), vp, vp, i64, vp, i32, vp, i64, vp],
        "mih_fit_mv": [vp, C.POINTER(_FitParams), vp, i64, vp, i64, vp, C.POINTER(_MvResult)],
        "mih_cv_mv": [vp, C.POINTER(_FitParams), vp, i64, vp, i64, vp, i32, vp, i64, i32, i32, vp],
        "mih_bench_xtv": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_float), C.POINTER(dbl)],
        "mih_xtv_algorithmic_bytes": [vp, C.c_int, C.POINTER(dbl)],
        "mih_abi_sizes": [vp, i32],
        "mih_session_create": [vp, C.POINTER(_FitParams), vp, vp, i64, vp, C.POINTER(vp)],
        "mih_session_step": [vp, C.POINTER(dbl), C.POINTER(i32), C.POINTER(dbl)],
        "mih_session_run": [vp, i64, C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl)],
        "mih_session_model": [vp, vp, vp],
        "mih_session_destroy": [vp],
        "mih_rccl_unique_id": [vp],
        "mih_comm_create_rccl": [vp, i32, i32, i32, i64, i64, C.POINTER(vp)],
        "mih_comm_destroy_rccl": [vp],
        "mih_comm_info": [vp, C.POINTER(i32), vp, i64],
        "mih_cv_allgather": [vp, vp, i64],
        "mih_profile_enable": [vp, C.c_int],
        "mih_profile_read": [vp, C.POINTER(dbl), C.POINTER(i64), C.c_int],
        "mih_profile_passes": [vp, C.POINTER(_PassRecord), i64, C.POINTER(i64), C.c_int],
        "mih_profile_counters": [vp, C.POINTER(i64), C.c_int],
        "mih_profile_exchange": [vp, C.POINTER(dbl), C.POINTER(i64), C.c_int],
    }
    if using_probes():       # include/mendeliht_hip_probes.h
        sig.update({"mih_probe_set_xtv_variant": [C.c_int], "mih_probe_set_xtv_multi_variant": [C.c_int],
                    "mih_probe_set_max_fused": [C.c_int],
                    "mih_probe_xtv_sequence": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]})
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    _lib = L
    return L


def exported_symbols():
    """Every symbol include/mendeliht_hip.h declares (checked by the CPU test-suite)."""
    return ["mih_device_count", "mih_last_error", "mih_version", "mih_snp_create", "mih_snp_create_synthetic",
            "mih_snp_create_synthetic_shard",
            "mih_dense_create", "mih_dense_create_synthetic", "mih_dense_create_f32", "mih_mat_destroy", "mih_mat_dims", "mih_mat_reserve",
            "mih_snp_mu_sigma", "mih_snp_export_bed", "mih_snp_naive_impute", "mih_xtv", "mih_xtv_batched", "mih_xv_sparse",
            "mih_project_topk", "mih_project_group_sparse", "mih_fit_iht", "mih_cv_iht", "mih_cv_meanloss", "mih_cv_assignment", "mih_cv_iht_multi", "mih_fit_iht_path",
            "mih_fit_mv", "mih_cv_mv", "mih_bench_xtv", "mih_xtv_algorithmic_bytes", "mih_xtv_batched_fmt", "mih_abi_sizes",
            "mih_session_create", "mih_session_step", "mih_session_run", "mih_session_model", "mih_session_destroy",
            "mih_rccl_unique_id", "mih_comm_create_rccl", "mih_comm_info", "mih_comm_destroy_rccl", "mih_cv_allgather",
            "mih_profile_enable", "mih_profile_read", "mih_profile_passes", "mih_profile_counters", "mih_profile_exchange"]


def probe_symbols():
    """What include/mendeliht_hip_probes.h declares: exported by the measurement build only."""
    return ["mih_probe_set_xtv_variant", "mih_probe_set_xtv_multi_variant", "mih_probe_set_max_fused", "mih_probe_xtv_sequence"]


def _check(rc):
    if rc == 0:
        return
    buf = C.create_string_buffer(512)
    lib().mih_last_error(buf, 512)
    raise _STATUS.get(rc, MendelIHTError)(f"[mih status {rc}] {buf.value.decode(errors='replace')}")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    _check(lib().mih_device_count(C.byref(n)))
    return n.value


# ---- distributions and links (Distributions.jl / GLM.jl names) -----------------------
class _Dist:
    code = 0
    name = "Normal"

    def __repr__(self):
        return f"{self.name}()"


class Normal(_Dist):
    code, name = 0, "Normal"


class Bernoulli(_Dist):
    code, name = 1, "Bernoulli"


class Poisson(_Dist):
    code, name = 2, "Poisson"


class NegativeBinomial(_Dist):
    code, name = 3, "NegativeBinomial"

    def __init__(self, r=1.0, p=0.5):
        self.r, self.p = float(r), float(p)

    def __repr__(self):
        return f"NegativeBinomial(r={self.r}, p={self.p})"


class Gamma(_Dist):
    """loglik_obs(::Gamma, ...) src/utilities.jl:34 (shape 1/phi, scale mu*phi); use with LogLink or InverseLink."""
    code, name = 4, "Gamma"


class InverseGaussian(_Dist):
    """loglik_obs(::InverseGaussian, ...) src/utilities.jl:35."""
    code, name = 5, "InverseGaussian"


class MvNormal(_Dist):
    code, name = -1, "MvNormal"


class IdentityLink:
    code = 0

    def __repr__(self):
        return "IdentityLink()"


class LogitLink:
    code = 1

    def __repr__(self):
        return "LogitLink()"


class LogLink:
    code = 2

    def __repr__(self):
        return "LogLink()"


def _link(name, code):
    return type(name, (), {"code": code, "__repr__": lambda self: f"{name}()"})


# the remaining GLM.jl Link types a caller may pass as `l` (linkinv / mueta closed forms in csrc/fit_common.h)
ProbitLink = _link("ProbitLink", 3)
CloglogLink = _link("CloglogLink", 4)
CauchitLink = _link("CauchitLink", 5)
InverseLink = _link("InverseLink", 6)
InverseSquareLink = _link("InverseSquareLink", 7)
SqrtLink = _link("SqrtLink", 8)


def canonicallink(d):
    """GLM.canonicallink (default `l` of cv_iht / iht_run_many_models, cross_validation.jl:237)."""
    d = _inst(d)
    return {0: IdentityLink, 1: LogitLink, 2: LogLink, 3: LogLink, 4: InverseLink, 5: InverseSquareLink}[max(d.code, 0)]()


def _inst(x):
    return x() if isinstance(x, type) else x


# ---- design matrices -------------------------------------------------------------------
def read_bed(path, n):
    """PLINK .bed (SNP-major, header 6c 1b 01) -> (p, ceil(n/4)) uint8 column bytes.  The file is memory-mapped
    (as SnpArrays.jl does): a 125 GB .bed is never copied into Python memory, its pages stream through the
    upload pipeline of mih_snp_create."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        magic = f.read(3)
    if size < 3 or magic != b"\x6c\x1b\x01":
        raise ArgumentError(f"{path} is not a SNP-major PLINK .bed file")
    stride = (n + 3) // 4
    if (size - 3) % stride:
        raise DimensionMismatch(f"{path}: size does not match n={n}")
    if size == 3:
        return np.zeros((0, stride), dtype=np.uint8)
    return np.memmap(path, dtype=np.uint8, mode="r", offset=3, shape=((size - 3) // stride, stride))


class _Mat:
    def __init__(self):
        self._h = C.c_void_p(None)
        self.n = self.p = 0

    def __del__(self):
        try:
            if self._h:
                lib().mih_mat_destroy(self._h)
                self._h = C.c_void_p(None)
        except Exception:
            pass

    @property
    def shape(self):
        return (self.n, self.p)

    def _dims(self):
        n, p = C.c_int64(0), C.c_int64(0)
        _check(lib().mih_mat_dims(self._h, C.byref(n), C.byref(p)))
        self.n, self.p = n.value, p.value

    # mul!(out, Transpose(x), r)
    def xtv(self, r, xtv_digits=None):
        r = np.asarray(r, dtype=np.float64)
        dg = _digits(xtv_digits)
        if r.ndim == 1:
            if r.size != self.n:
                raise DimensionMismatch(f"r has length {r.size}, expected {self.n}")
            r = np.ascontiguousarray(r)
            out = np.empty(self.p)
            if dg == 0:
                _check(lib().mih_xtv(self._h, _p(r), _p(out)))
            else:
                _check(lib().mih_xtv_batched_fmt(self._h, _p(r), 1, dg, _p(out)))
            return out
        if r.shape[0] != self.n:
            raise DimensionMismatch(f"R has {r.shape[0]} rows, expected {self.n}")
        R = np.asfortranarray(r)
        out = np.empty((self.p, R.shape[1]), order="F")
        _check(lib().mih_xtv_batched_fmt(self._h, _p(R), R.shape[1], dg, _p(out)))
        return out

    def xv_sparse(self, idx, val):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.float64)
        out = np.empty(self.n)
        _check(lib().mih_xv_sparse(self._h, _p(idx), _p(val), idx.size, _p(out)))
        return out

    def bench_xtv(self, variant=-1, iters=10, warmup=2, seed=1, xtv_digits=None):
        """ms per single-residual pass (the workspace of a single fit) and a checksum of the result."""
        return self.bench_xtv_batched(1, variant=variant, iters=iters, warmup=warmup, seed=seed, xtv_digits=xtv_digits)

    def bench_xtv_batched(self, m, max_fused=4, variant=-1, iters=5, warmup=1, seed=1, xtv_digits=None):
        """`variant` / `max_fused` other than the defaults are knobs of the measurement build (MENDELIHT_HIP_PROBES=1)."""
        if variant != -1 or max_fused != 4 or using_probes():
            probe_set(variant=variant, max_fused=max_fused)
        ms, cs = C.c_float(0), C.c_double(0)
        try:
            _check(lib().mih_bench_xtv(self._h, _digits(xtv_digits), m, iters, warmup, seed, C.byref(ms), C.byref(cs)))
        finally:
            if using_probes():
                probe_set(variant=-1, max_fused=4)
        return ms.value, cs.value

    def algorithmic_bytes(self, m=1):
        b = C.c_double(0)
        _check(lib().mih_xtv_algorithmic_bytes(self._h, m, C.byref(b)))
        return b.value


# The `reserve` argument of SnpLinAlg when the caller leaves it out.  False: the library's own policy (a 2-bit matrix of 4 GiB or
# more keeps a reserve of device memory for its fits, smaller ones allocate per fit).  True: every matrix asks for a reserve
# (mih_mat_reserve) -- tests/conftest.py sets it so that small test matrices run the pool / arena code of a 125 GB matrix.
RESERVE_BY_DEFAULT = False


class SnpLinAlg(_Mat):
    """SnpLinAlg{Float64}(s::SnpArray; model=ADDITIVE_MODEL, center, scale, impute) on the GPU."""

    def _reserve(self, reserve):
        if reserve:
            _check(lib().mih_mat_reserve(self._h, 0))
        elif reserve is None and RESERVE_BY_DEFAULT:
            lib().mih_mat_reserve(self._h, 0)          # best effort: a crowded device keeps the per-fit allocations

    def __init__(self, bed, n=None, center=False, scale=False, impute=True, device=0, _handle=None, dtype=np.float64, reserve=None):
        """dtype: the element type T of SnpLinAlg{T} (np.float64 or np.float32, src/MendelIHT.jl:39).  The device arithmetic is
        the same for both (exact fixed point + Float64); with float32 the models fit_iht / cv_iht return are cast to float32.
        reserve: True asks for the reserve of device memory a large matrix keeps for its fits (mih_mat_reserve), None = RESERVE_BY_DEFAULT."""
        super().__init__()
        self.center, self.scale, self.impute = bool(center), bool(scale), bool(impute)
        self.device = device
        self.dtype = np.dtype(dtype).type
        if self.dtype not in (np.float64, np.float32):
            raise ArgumentError("SnpLinAlg{T}: T must be Float64 or Float32")
        if _handle is not None:
            self._h = _handle
            self._dims()
            return
        if isinstance(bed, (str, os.PathLike)):
            if n is None:
                raise ArgumentError("n (number of samples) is required with a .bed path")
            bed = read_bed(bed, n)
        cols = np.ascontiguousarray(bed, dtype=np.uint8)
        if cols.ndim != 2:
            raise DimensionMismatch("bed columns must be a (p, stride) uint8 array")
        if n is None:
            raise ArgumentError("n (number of samples) is required")
        h = C.c_void_p(None)
        _check(lib().mih_snp_create(_p(cols), n, cols.shape[0], cols.shape[1], int(center), int(scale),
                                    int(impute), 32 if self.dtype is np.float32 else 64, device, C.byref(h)))
        self._h = h
        self._dims()
        self._reserve(reserve)

    @classmethod
    def synthetic(cls, n, p, seed=2024, missing_rate=0.0, center=True, scale=True, impute=True, device=0,
                  col_offset=0, reserve=None):
        """Columns [col_offset, col_offset + p) of the seeded synthetic SnpArray (col_offset > 0: one shard)."""
        h = C.c_void_p(None)
        _check(lib().mih_snp_create_synthetic_shard(n, p, int(col_offset), seed, float(missing_rate), int(center),
                                                    int(scale), int(impute), device, C.byref(h)))
        x = cls(None, center=center, scale=scale, impute=impute, device=device, _handle=h)
        x._reserve(reserve)
        return x

    def mu_sigma(self):
        mu, s = np.empty(self.p), np.empty(self.p)
        _check(lib().mih_snp_mu_sigma(self._h, _p(mu), _p(s)))
        return mu, s

    def export_bed(self):
        out = np.empty((self.p, (self.n + 3) // 4), dtype=np.uint8)
        _check(lib().mih_snp_export_bed(self._h, _p(out)))
        return out


def naive_impute(x, destination, n=None):
    """naive_impute(x::SnpArray, destination) -- src/utilities.jl:862-899: writes `destination` (.bed) with every missing
    genotype of `x` replaced by the mode of its SNP.  `x` is a PLINK .bed path or the (p, ceil(n/4)) column bytes (then
    `n` is required), or a SnpLinAlg already on the GPU."""
    if isinstance(x, SnpLinAlg):
        mat = x
    else:
        if n is None:
            raise ArgumentError("naive_impute needs the number of samples n with a .bed path or raw columns")
        cols = read_bed(x, n) if isinstance(x, (str, os.PathLike)) else x
        mat = SnpLinAlg(cols, n)
    out = np.empty((mat.p, (mat.n + 3) // 4), dtype=np.uint8)
    _check(lib().mih_snp_naive_impute(mat._h, _p(out)))
    if not str(destination).endswith(".bed"):
        destination = str(destination) + ".bed"
    with open(destination, "wb") as f:
        f.write(b"\x6c\x1b\x01")
        f.write(out.tobytes())
    return None


class DenseMatrix(_Mat):
    """The reference's `x::Matrix{Float64}` / `x::Matrix{Float32}` design matrix, resident in HBM.  A float32 input
    is stored as Float32 on the device (half the memory traffic); all arithmetic stays Float64."""

    def __init__(self, x, device=0, _handle=None):
        super().__init__()
        self.device = device
        self.dtype = np.float64
        if _handle is not None:
            self._h = _handle
            self._dims()
            return
        x = np.asarray(x)
        if x.ndim != 2:
            raise DimensionMismatch("x must be a matrix")
        h = C.c_void_p(None)
        if x.dtype == np.float32:
            x = np.asfortranarray(x)
            self.dtype = np.float32
            _check(lib().mih_dense_create_f32(_p(x), x.shape[0], x.shape[1], device, C.byref(h)))
        else:
            x = np.asfortranarray(x, dtype=np.float64)
            _check(lib().mih_dense_create(_p(x), x.shape[0], x.shape[1], device, C.byref(h)))
        self._h = h
        self._dims()

    @classmethod
    def synthetic(cls, n, p, seed=2024, device=0):
        h = C.c_void_p(None)
        _check(lib().mih_dense_create_synthetic(n, p, seed, device, C.byref(h)))
        return cls(None, device=device, _handle=h)


def _as_mat(x):
    if isinstance(x, _Mat):
        return x
    if isinstance(x, np.ndarray):
        if x.dtype == np.uint8:
            raise ArgumentError("x is a SnpArray! Please convert it to a SnpLinAlg first!")
        return DenseMatrix(x)
    raise ArgumentError(f"unsupported design matrix type {type(x)}")


# ---- projections ---------------------------------------------------------------------------
def project_k(x, k):
    """project_k!(x, k): keep the k largest |x_i| (ties kept); returns the projected copy."""
    if k < 0:
        raise ArgumentError(f"DomainError: Attempted to project to sparsity level {k}")
    x = np.array(x, dtype=np.float64).ravel()
    kept = C.c_int64(0)
    _check(lib().mih_project_topk(_p(x), x.size, int(k), C.byref(kept)))
    return x


def project_group_sparse(y, group, J, k):
    y = np.array(y, dtype=np.float64).ravel()
    group = np.ascontiguousarray(group, dtype=np.int64)
    if group.size != y.size:
        raise DimensionMismatch("group must have the length of y")
    kv = np.ascontiguousarray(np.atleast_1d(k), dtype=np.int64)
    _check(lib().mih_project_group_sparse(_p(y), _p(group), y.size, int(J), _p(kv), int(np.ndim(k) > 0)))
    return y


def standardize(z):
    """standardize!(z) (src/utilities.jl:494-530): column-wise (z - mean) * 1/sample-sd."""
    z = np.array(z, dtype=np.float64)
    mu = z.mean(axis=0)
    sd = np.sqrt(((z - mu) ** 2).sum(axis=0) / (z.shape[0] - 1))
    return (z - mu) / sd


# ---- results -------------------------------------------------------------------------------
class IHTResult:
    """IHTResult (src/data_structures.jl:245-256)."""

    def __init__(self, time, logl, iter, beta, c, J, k, group, d, sigma_g, trace=None, choose_fired=False, mu=None):
        self.time, self.logl, self.iter = time, logl, iter
        self.beta, self.c, self.J, self.k, self.group, self.d = beta, c, J, k, group, d
        self.σg = self.sigma_g = sigma_g
        self.trace = trace or {}
        self.choose_fired = choose_fired
        self.mu = mu

    def __repr__(self):
        nz = np.flatnonzero(self.beta)
        cz = np.flatnonzero(self.c)
        lines = ["", f"IHT estimated {nz.size} nonzero SNP predictors and {cz.size} non-genetic predictors.", "",
                 f"Compute time (sec):     {self.time}", f"Final loglikelihood:    {self.logl}",
                 f"SNP PVE:                {self.σg}", f"Iterations:             {self.iter}", "",
                 "Selected genetic predictors:", " Position  Estimated_β"]
        lines += [f" {j + 1:8d}  {self.beta[j]: .6g}" for j in nz]
        lines += ["", "Selected nongenetic predictors:", " Position  Estimated_β"]
        lines += [f" {j + 1:8d}  {self.c[j]: .6g}" for j in cz]
        return "\n".join(lines)


class mIHTResult:
    """mIHTResult (src/data_structures.jl:263-273)."""

    def __init__(self, time, logl, iter, beta, c, k, traits, Sigma, sigma_g, trace=None, choose_fired=False):
        self.time, self.logl, self.iter, self.beta, self.c = time, logl, iter, beta, c
        self.k, self.traits, self.Σ, self.σg = k, traits, Sigma, sigma_g
        self.Sigma, self.sigma_g = Sigma, sigma_g
        self.trace = trace or {}
        self.choose_fired = choose_fired


class IHTSession:
    """An IHTVariable kept alive on the GPU: `initialize` once, then `step()` = one iht_one_step!."""

    def __init__(self, y, x, z=None, *, k=10, J=1, d=None, l=None, zkeep=None, weight=None, max_step=3, train=None,
                 comm=None, xtv_digits=None, step_mode=None):
        x = _as_mat(x)
        d = _inst(d) if d is not None else Normal()
        l = _inst(l) if l is not None else IdentityLink()
        self.x = x
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).ravel())
        z = np.ones((x.n, 1)) if z is None else np.asarray(z, dtype=np.float64)
        z = np.asfortranarray(z.reshape(z.shape[0], -1))
        if not (y.size == x.n == z.shape[0]):
            raise DimensionMismatch(f"row dimension of y, x, and z ({y.size}, {x.n}, {z.shape[0]}) are not equal")
        self.q = z.shape[1]
        self._keep = [y, z]
        prm = _params(k, J, d, l, 1e-4, 1 << 30, 5, max_step, "None", zkeep, weight, None, self.q, x.p, self._keep,
                      comm=comm, xtv_digits=xtv_digits, step_mode=step_mode)
        tr = None if train is None else np.ascontiguousarray(train, dtype=np.uint8)
        self._h = C.c_void_p(None)
        _check(lib().mih_session_create(x._h, C.byref(prm), _p(y), _p(z), self.q, _p(tr), C.byref(self._h)))

    def step(self):
        logl, bt, tol = C.c_double(0), C.c_int32(0), C.c_double(0)
        _check(lib().mih_session_step(self._h, C.byref(logl), C.byref(bt), C.byref(tol)))
        return logl.value, bt.value, tol.value

    def run(self, nsteps):
        """`nsteps` iterations in one library call; returns (logl, total backtracks, tol) of the last one."""
        logl, bt, tol = C.c_double(0), C.c_int64(0), C.c_double(0)
        _check(lib().mih_session_run(self._h, int(nsteps), C.byref(logl), C.byref(bt), C.byref(tol)))
        return logl.value, bt.value, tol.value

    def model(self):
        beta, c = np.zeros(self.x.p), np.zeros(self.q)
        _check(lib().mih_session_model(self._h, _p(beta), _p(c)))
        return beta, c

    def close(self):
        if self._h:
            lib().mih_session_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_XTV_FORMATS = (0, -1, 4910, 4908, 1316, 1308, 428)
_default_digits = 0


def set_xtv_digits(digits=0):
    """Default of the `xtv_digits=` keyword of this mirror's calls (fit_iht, cv_iht, iht_run_many_models, IHTSession,
    SnpLinAlg.xtv, ...): the fixed-point format of the residual in X'r, id = base * 100 + digits -- 0 = library default =
    4910 (10 base-49 FP6 digits, 54-bit, 19 residuals in the six operands of a full pass) for fused passes and 428 (28 base-4 digits) for
    a single fit; 1316 (16 base-13 FP4 digits, 57-bit, two per operand); 4908 (43-bit, four per operand: the opt-in fast
    mode for fused multi-RHS passes); 1308 (27-bit, four per operand).  The LIBRARY has no such global: the format travels
    with every call (mih_fit_params::xtv_digits, mih_xtv_batched_fmt), so concurrent calls may differ."""
    global _default_digits
    if int(digits) not in _XTV_FORMATS:
        raise ArgumentError("residual format must be 0 (default), -1 (auto), 4910, 4908, 1316, 1308 or 428")
    _default_digits = int(digits)


def _digits(xtv_digits=None):
    d = _default_digits if xtv_digits is None else int(xtv_digits)
    if d not in _XTV_FORMATS:
        raise ArgumentError("residual format must be 0 (default), -1 (auto), 4910, 4908, 1316, 1308 or 428")
    return d


def probe_set(variant=None, multi_variant=None, max_fused=None):
    """Knobs of the measurement build (libmendeliht_hip_probes.so, MENDELIHT_HIP_PROBES=1): the per-wave single-operand
    kernel shapes, the launch shape / probe id of the LDS-shared and ring kernels, operands fused per register-staged pass."""
    if not using_probes():
        if (variant in (None, -1)) and (multi_variant in (None, 0)) and (max_fused in (None, 4)):
            return
        raise MendelIHTError("kernel-shape knobs exist in the measurement build only: set MENDELIHT_HIP_PROBES=1 before "
                             "the first call (the product library has one kernel per format and operand count)")
    L = lib()
    if variant is not None:
        _check(L.mih_probe_set_xtv_variant(int(variant)))
    if multi_variant is not None:
        _check(L.mih_probe_set_xtv_multi_variant(int(multi_variant)))
    if max_fused is not None:
        _check(L.mih_probe_set_max_fused(int(max_fused)))


def profile_enable(x, on=True):
    """Measurement hook of matrix `x` (mih_profile_*): record every launch of the dominant X'r kernel on it."""
    _check(lib().mih_profile_enable(_as_mat(x)._h, int(on)))


def profile_read(x, reset=True):
    """(total kernel ms, launches) since the last reset."""
    ms, n = C.c_double(0), C.c_int64(0)
    _check(lib().mih_profile_read(_as_mat(x)._h, C.byref(ms), C.byref(n), int(reset)))
    return ms.value, n.value


def profile_passes(x, reset=True):
    """The recorded launches, oldest first: dicts with kernel, residuals, operands, stream_tag, start_ms, ms."""
    h = _as_mat(x)._h
    n = C.c_int64(0)
    _check(lib().mih_profile_passes(h, None, 0, C.byref(n), 0))
    buf = (_PassRecord * max(n.value, 1))()
    _check(lib().mih_profile_passes(h, buf, n.value, C.byref(n), int(reset)))
    return [dict(kernel=buf[i].kernel.decode(), residuals=buf[i].residuals, operands=buf[i].operands,
                 stream_tag=buf[i].stream_tag, start_ms=buf[i].start_ms, ms=buf[i].ms) for i in range(n.value)]


def profile_counters(x, reset=True):
    """What the lock-step drivers did on `x` while its hook was on (PROFILE_COUNTERS)."""
    out = (C.c_int64 * len(PROFILE_COUNTERS))()
    _check(lib().mih_profile_counters(_as_mat(x)._h, out, int(reset)))
    return dict(zip(PROFILE_COUNTERS, [int(v) for v in out]))


EXCHANGE_KINDS = ("allreduce_n_plus_1", "allreduce_n", "allgather_candidates", "host_scalars")


def profile_exchange(x, reset=True):
    """Exchanges of the column-sharded fits on `x` while its hook was on (mih_profile_exchange): {kind: (count, summed ms)}."""
    ms, cnt = (C.c_double * 4)(), (C.c_int64 * 4)()
    _check(lib().mih_profile_exchange(_as_mat(x)._h, ms, cnt, int(reset)))
    return {kk: dict(count=int(cnt[i]), ms=float(ms[i])) for i, kk in enumerate(EXCHANGE_KINDS)}


def busy_union_ms(passes):
    """Time during which at least one of the recorded launches was running (launches of different lanes overlap)."""
    iv = sorted((q["start_ms"], q["start_ms"] + q["ms"]) for q in passes)
    tot, cur_a, cur_b = 0.0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def cv_assignment(path, q, world):
    """rank_of[fold, ik] of the library's sharding rule for cv_iht (mih_cv_assignment): a (q, len(path)) int32 array."""
    path = np.ascontiguousarray(list(path), dtype=np.int64)
    out = np.zeros((int(q), path.size), dtype=np.int32)
    _check(lib().mih_cv_assignment(_p(path), path.size, int(q), int(world), _p(out)))
    return out


def hash_folds(n, q, seed=2026):
    """folds_i = 1 + (hash(seed, i) mod q): explicit, RNG-free fold labels for benchmarks and tests (the
    reference's default is rand(1:q, n), cross_validation.jl:72)."""
    i = np.arange(n, dtype=np.uint64)
    x = (i + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(31)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(29)
    return (1 + (x % np.uint64(q))).astype(np.int32)


def _is_multivariate(y):
    y = np.asarray(y)
    return y.ndim == 2 and y.shape[0] > 1 and y.shape[1] > 1


def _print_signature(io):
    print("****                   MendelIHT (mendeliht.jl_amd, MI355X)         ****", file=io)
    print("****     hot path of OpenMendel/MendelIHT.jl v1.4.11 on gfx950      ****", file=io)
    print("", file=io)


def _print_parameters(io, k, d, l, use_maf, group, debias, tol, max_iter, min_iter):
    reg = {"Normal": "linear", "Bernoulli": "logistic", "Poisson": "Poisson", "NegativeBinomial": "NegativeBinomial",
           "MvNormal": "Multivariate Gaussian"}.get(d.name, "unknown")
    print(f"Running sparse {reg} regression", file=io)
    print(f"Link functin = {l}", file=io)
    if np.ndim(k) == 0:
        print(f"Sparsity parameter (k) = {k}", file=io)
    else:
        print("Sparsity parameter (k) = using group membership specified in k", file=io)
    print(f"Prior weight scaling = {'on' if use_maf else 'off'}", file=io)
    print(f"Doubly sparse projection = {'on' if group is not None and len(group) > 0 else 'off'}", file=io)
    print(f"Debias = {'on' if debias else 'off'}", file=io)
    print(f"Max IHT iterations = {max_iter}", file=io)
    print(f"Converging when tol < {tol} and iteration ≥ {min_iter}:\n", file=io)


_default_step_mode = 0


def set_step_mode(mode=0):
    """Default of the `step_mode=` keyword (mih_fit_params::step_mode): 0 = iht_one_step! resident on the device wherever the fit
    allows it, 1 = every step host-driven (the path of rounds 1-4).  Same results bit for bit; the switch exists for A/B runs."""
    global _default_step_mode
    if int(mode) not in (0, 1):
        raise ArgumentError("step_mode must be 0 (device-resident steps) or 1 (host-driven steps)")
    _default_step_mode = int(mode)


def _params(k, J, d, l, tol, max_iter, min_iter, max_step, est_r, zkeep, weight, group, q, p, keep, progress=None,
            init_beta=False, comm=None, debias=False, xtv_digits=None, choose=None, step_mode=None):
    prm = _FitParams()
    prm.step_mode = _default_step_mode if step_mode is None else int(step_mode)
    if choose is not None:
        ccb = _choose_callback(choose)
        prm.choose = C.cast(ccb, C.c_void_p)
        keep.append(ccb)
    prm.debias = int(bool(debias))
    prm.xtv_digits = _digits(xtv_digits)
    if comm is not None:            # column-sharded fit: mendeliht.jl_amd.dist.ColumnComm
        prm.comm = comm.pointer()
        keep.append(comm)
    ks = None
    if np.ndim(k) > 0:
        ks = np.ascontiguousarray(k, dtype=np.int64)
        if group is None or len(group) <= 1:
            raise ArgumentError("Doubly sparse projection specified (since k is a vector) but there are no group information.")
        prm.k = 0
    else:
        if k < 0:
            raise ArgumentError("Value of k (max predictors per group) must be nonnegative!")
        prm.k = int(k)
    prm.J = int(J)
    prm.dist = max(d.code, 0)
    prm.link = l.code
    prm.nb_r = getattr(d, "r", 1.0)
    prm.tol = float(tol)
    prm.max_iter, prm.min_iter, prm.max_step = int(max_iter), int(min_iter), int(max_step)
    er = est_r if isinstance(est_r, str) else ("None" if est_r is None else str(est_r))
    er = er.lstrip(":").lower()
    if er not in ("none", "mm", "newton"):
        raise ArgumentError(f"Only support method is Newton or MM, but got {est_r}")
    prm.est_r = {"none": 0, "mm": 1, "newton": 2}[er]
    zk = None
    if zkeep is not None:
        zk = np.ascontiguousarray(zkeep, dtype=np.uint8)
        if zk.size != q:
            raise DimensionMismatch(f"zkeep must have length {q} but was {zk.size}")
    w = None
    if weight is not None and len(weight) > 0:
        w = np.ascontiguousarray(weight, dtype=np.float64)
        if w.size != p:
            raise DimensionMismatch(f"weight must have length {p} but was {w.size}")
    g = None
    if group is not None and len(group) > 0:
        g = np.ascontiguousarray(group, dtype=np.int64)
        if g.size != p:
            raise DimensionMismatch(f"group must have length {p} but was {g.size}")
    prm.zkeep, prm.weight, prm.group, prm.ks = _p(zk), _p(w), _p(g), _p(ks)
    prm.nks = 0 if ks is None else ks.size
    prm.init_beta = int(bool(init_beta))
    cb = None
    if progress is not None:
        cb = _PROGRESS(progress)
        prm.progress = C.cast(cb, C.c_void_p)
    keep.extend([zk, w, g, ks, cb])
    return prm


def fit_iht(y, x, z=None, *, k=10, J=1, d=None, l=None, group=None, weight=None, zkeep=None, est_r="None",
            use_maf=False, debias=False, verbose=True, tol=1e-4, max_iter=200, min_iter=5, max_step=3,
            io=None, init_beta=False, memory_efficient=True, train=None, comm=None, xtv_digits=None, choose=None, step_mode=None):
    """fit_iht(y, x, z; k, J, d, l, ...) -- src/fit.jl:60-118.

    step_mode (no reference counterpart): how iht_one_step! is driven (set_step_mode); None = the mirror's default.

    choose (no keyword in the reference, which draws from the global RNG): fn(kind, list, excess) making the random draw of
    _choose! (src/utilities.jl:444-458, src/multivariate.jl:310-351) when a projection leaves exact ties -- see
    _choose_callback; None = the library's deterministic rule, flagged by result.choose_fired.

    xtv_digits (no reference counterpart): fixed-point format of the residual in this call's X'r passes (set_xtv_digits).

    comm: a `dist.ColumnComm` when x holds only this process's block of SNP columns (column-sharded fit
    over several GPUs; beta in the result then covers the local columns -- see dist.fit_iht_sharded).

    Univariate: y (n,), x SnpLinAlg/DenseMatrix (n x p), z (n, q) with a leading column of ones.
    Multivariate (d=MvNormal or y 2-D): y (r, n), z (q, n), x is the same SnpLinAlg (its transpose is implied).
    """
    io = io or sys.stdout
    x = _as_mat(x)
    mv = _is_multivariate(y)
    d = _inst(d) if d is not None else (MvNormal() if mv else Normal())
    l = _inst(l) if l is not None else IdentityLink()
    if debias and mv:
        raise ArgumentError("debias is disabled for multivariate traits (multivariate.jl:569-570)")
    if init_beta and not isinstance(d, (Normal, MvNormal)):
        raise ArgumentError("Intializing beta values only work for Gaussian phenotypes! Sorry!")
    if not memory_efficient:
        raise ArgumentError("the GPU path is always memory_efficient=true")
    if isinstance(x, SnpLinAlg):
        if not x.center:
            raise ArgumentError("x is not centered! Please construct SnpLinAlg{Float64}(::SnpArray, center=true, scale=true)")
        if not x.scale:
            print("Warning: x is not scaled! We highly recommend `scale=true` in `SnpLinAlg` constructor", file=sys.stderr)
        if not x.impute:
            print("Warning: x does not have impute flag! We highly recommend `impute=true` in `SnpLinAlg` constructor", file=sys.stderr)
    if verbose:
        _print_signature(io)
    if mv:
        return _fit_mv(y, x, z, k, d, l, zkeep, verbose, tol, max_iter, min_iter, max_step, io, train, init_beta, xtv_digits, choose, comm)
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).ravel())
    n = x.n
    z = np.ones((n, 1)) if z is None else np.asarray(z, dtype=np.float64)
    z = np.asfortranarray(z.reshape(z.shape[0], -1))
    if not (y.size == n == z.shape[0]):
        raise DimensionMismatch(f"row dimension of y, x, and z ({y.size}, {n}, {z.shape[0]}) are not equal")
    _checky(y, d)
    q = z.shape[1]
    lines = []

    def progress(_user, it, logl, bt, tl):
        line = f"Iteration {it}: loglikelihood = {logl!r}, backtracks = {bt}, tol = {tl!r}"
        lines.append(line)
        print(line, file=io)

    # the per-iteration callback only when its lines are printed as they come (verbose): a fit whose steps the HOST drives waits
    # for it between two steps (tens of microseconds of interpreter per iteration); a quiet fit gets the same lines from the trace
    keep = []
    prm = _params(k, J, d, l, tol, max_iter, min_iter, max_step, est_r, zkeep, weight, group, q, x.p, keep, progress if verbose else None,
                  init_beta=init_beta, comm=comm, debias=debias, xtv_digits=xtv_digits, choose=choose, step_mode=step_mode)
    if verbose:
        _print_parameters(io, k, d, l, use_maf, group, debias, tol, max_iter, min_iter)
    tr = None if train is None else np.ascontiguousarray(train, dtype=np.uint8)
    beta, c, mu = np.zeros(x.p), np.zeros(q), np.zeros(n)
    nt = max(int(max_iter), 1)
    lt, tt, bt = np.zeros(nt), np.zeros(nt), np.zeros(nt, dtype=np.int32)
    res = _FitResult()
    res.beta, res.c, res.mu = _p(beta), _p(c), _p(mu)
    res.logl_trace, res.tol_trace, res.bt_trace = _p(lt), _p(tt), _p(bt)
    _check(lib().mih_fit_iht(x._h, C.byref(prm), _p(y), _p(z), q, _p(tr), C.byref(res)))
    m = res.n_trace
    if not verbose:
        lines = _trace_lines(lt, bt, tt, m)
    if verbose and res.iter >= max_iter:
        print(f"Did not converge after {max_iter} iterations! IHT run time was {res.time} seconds", file=io)
    dd = NegativeBinomial(res.nb_r) if isinstance(d, NegativeBinomial) else d
    if getattr(x, "dtype", np.float64) is np.float32:          # SnpLinAlg{Float32}: the model comes back in the caller's T
        beta, c, mu = beta.astype(np.float32), c.astype(np.float32), mu.astype(np.float32)
    return IHTResult(res.time, res.logl, res.iter, beta, c, J, k, np.array([] if group is None else group), dd, res.pve,
                     trace=dict(logl=lt[:m].copy(), tol=tt[:m].copy(), backtracks=bt[:m].copy(), lines=lines),
                     choose_fired=bool(res.choose_fired), mu=mu)


def _trace_lines(lt, bt, tt, m):
    """the lines the progress callback prints, from the trace arrays of a quiet fit"""
    return [f"Iteration {i + 1}: loglikelihood = {float(lt[i])!r}, backtracks = {int(bt[i])}, tol = {float(tt[i])!r}" for i in range(m)]


def _checky(y, d):
    if isinstance(d, Bernoulli) and not np.all((y == 0) | (y == 1)):
        raise ArgumentError("Bernoulli data y must be 1 or 0 only")
    if isinstance(d, (Poisson, NegativeBinomial)) and (np.any(y < 0) or np.any(y != np.floor(y))):
        raise ArgumentError("Poisson/NegativeBinomial data must be nonnegative integers")
    if isinstance(d, (Gamma, InverseGaussian)) and np.any(y <= 0):
        raise ArgumentError("Gamma/InverseGaussian data must be positive")


def _fit_mv(Y, x, Z, k, d, l, zkeep, verbose, tol, max_iter, min_iter, max_step, io, train, init_beta=False, xtv_digits=None,
            choose=None, comm=None):
    Y = np.asfortranarray(np.asarray(Y, dtype=np.float64))
    r, n = Y.shape
    Z = np.ones((1, n)) if Z is None else np.asarray(Z, dtype=np.float64)
    Z = np.asfortranarray(Z.reshape(-1, Z.shape[-1]) if Z.ndim > 1 else Z.reshape(1, -1))
    if not (n == x.n == Z.shape[1]):
        raise DimensionMismatch(f"number of samples in y, x, and z = {n}, {x.n}, {Z.shape[1]} are not equal")
    if np.ndim(k) > 0:
        raise ArgumentError("multivariate IHT takes an integer k")
    q = Z.shape[0]
    lines = []

    def progress(_user, it, logl, bt, tl):
        line = f"Iteration {it}: loglikelihood = {logl!r}, backtracks = {bt}, tol = {tl!r}"
        lines.append(line)
        print(line, file=io)

    keep = []
    prm = _params(k, 1, Normal(), l, tol, max_iter, min_iter, max_step, "None", zkeep, None, None, q, x.p, keep, progress if verbose else None,
                  init_beta=init_beta, xtv_digits=xtv_digits, choose=choose, comm=comm)
    if verbose:
        _print_parameters(io, k, MvNormal(), l, False, None, False, tol, max_iter, min_iter)
    tr = None if train is None else np.ascontiguousarray(train, dtype=np.uint8)
    B, Cm = np.zeros((r, x.p), order="F"), np.zeros((r, q), order="F")
    S, pve = np.zeros((r, r), order="F"), np.zeros(r)
    nt = max(int(max_iter), 1)
    lt, tt, bt = np.zeros(nt), np.zeros(nt), np.zeros(nt, dtype=np.int32)
    res = _MvResult()
    res.B, res.C, res.Sigma, res.pve = _p(B), _p(Cm), _p(S), _p(pve)
    res.logl_trace, res.tol_trace, res.bt_trace = _p(lt), _p(tt), _p(bt)
    _check(lib().mih_fit_mv(x._h, C.byref(prm), _p(Y), r, _p(Z), q, _p(tr), C.byref(res)))
    m = res.n_trace
    if not verbose:
        lines = _trace_lines(lt, bt, tt, m)
    return mIHTResult(res.time, res.logl, res.iter, B, Cm, k, r, S, pve,
                      trace=dict(logl=lt[:m].copy(), tol=tt[:m].copy(), backtracks=bt[:m].copy(), lines=lines),
                      choose_fired=bool(res.choose_fired))


def cv_iht(y, x, z=None, *, d=None, l=None, path=range(1, 21), q=5, est_r="None", group=None, weight=None,
           zkeep=None, folds=None, debias=False, verbose=True, max_iter=100, min_iter=5, init_beta=False,
           memory_efficient=True, tol=1e-4, max_step=3, rank=0, world=1, reduce=None, return_raw=False, xtv_digits=None,
           cv_threads=0):
    """cv_iht(y, x, z; path, q, folds, ...) -- src/cross_validation.jl:60-131.

    `x` may be a list of replicas of the matrix (one per GPU): the combinations are then spread over them
    from this one process (mih_cv_iht_multi), as the reference spreads them over its threads.
    `rank`/`world` shard the (fold, k) combinations over processes (one GPU each); `reduce`
    is a callable that sum-reduces the raw q x len(path) loss matrix across ranks (see
    mendeliht.jl_amd.dist.cv_iht_distributed for the torch.distributed/RCCL version).
    `cv_threads` (est_r only): the reference re-uses one IHTVariable per Julia thread, so the NegBin r of one fit is the starting
    value of that thread's next fit (cross_validation.jl:91,100-110) and its losses depend on Threads.nthreads(); the library
    follows the chains of `cv_threads` threads in lock-step.  0 (the default) = 1 = one chain over the whole grid = the reference at
    its default Threads.nthreads() == 1 -- the same meaning in mih_fit_params, the CPU checker of the tests and the Julia glue (which passes
    Threads.nthreads()); cv_threads=q gives one chain per fold (faster, but not a default reference run's losses).
    """
    replicas = None
    if isinstance(x, (list, tuple)):                 # one replica of the matrix per GPU, driven from this process
        replicas = [_as_mat(r) for r in x]
        if not replicas:
            raise ArgumentError("empty list of matrix replicas")
        x = replicas[0]
        if world != 1:
            raise ArgumentError("pass either a list of replicas (one process, several GPUs) or rank/world (one process per GPU)")
    x = _as_mat(x)
    mv = _is_multivariate(y)
    d = _inst(d) if d is not None else (MvNormal() if mv else Normal())
    l = _inst(l) if l is not None else IdentityLink()
    if not memory_efficient:
        raise ArgumentError("the GPU path is always memory_efficient=true")
    if debias and mv:
        raise ArgumentError("debias is disabled for multivariate traits (multivariate.jl:569-570)")
    if replicas is not None and mv:
        raise ArgumentError("replica lists are supported for univariate cross-validation")
    if init_beta and not isinstance(d, (Normal, MvNormal)):
        raise ArgumentError("Intializing beta values only work for Gaussian phenotypes! Sorry!")
    path = np.ascontiguousarray(list(path), dtype=np.int64)
    n = x.n
    if path.size == 0:
        raise ArgumentError("path is empty")
    if path.max() > x.p:
        raise ArgumentError("Sparsity level in `path` cannot be larger than total number of variables")
    if folds is None:
        folds = np.random.randint(1, q + 1, size=n)      # rand(1:q, n) (cross_validation.jl:72)
    folds = np.ascontiguousarray(folds, dtype=np.int32)
    if folds.size != n:
        raise DimensionMismatch("folds must have one label per sample")
    raw = np.zeros((q, path.size))
    keep = []
    if mv:
        Y = np.asfortranarray(np.asarray(y, dtype=np.float64))
        r = Y.shape[0]
        Z = np.ones((1, n)) if z is None else np.asarray(z, dtype=np.float64)
        Z = np.asfortranarray(Z.reshape(-1, Z.shape[-1]) if Z.ndim > 1 else Z.reshape(1, -1))
        prm = _params(1, 1, Normal(), l, tol, max_iter, min_iter, max_step, "None", zkeep, None, None, Z.shape[0], x.p, keep,
                      init_beta=init_beta, xtv_digits=xtv_digits)
        _check(lib().mih_cv_mv(x._h, C.byref(prm), _p(Y), r, _p(Z), Z.shape[0], _p(folds), q, _p(path), path.size,
                               rank, world, _p(raw)))
    else:
        yv = np.ascontiguousarray(np.asarray(y, dtype=np.float64).ravel())
        zz = np.ones((n, 1)) if z is None else np.asarray(z, dtype=np.float64)
        zz = np.asfortranarray(zz.reshape(zz.shape[0], -1))
        if not (yv.size == n == zz.shape[0]):
            raise DimensionMismatch(f"row dimension of y, x, and z ({yv.size}, {n}, {zz.shape[0]}) are not equal")
        _checky(yv, d)
        prm = _params(1, 1, d, l, tol, max_iter, min_iter, max_step, est_r, zkeep, weight, group, zz.shape[1], x.p, keep,
                      init_beta=init_beta, debias=debias, xtv_digits=xtv_digits)
        prm.cv_threads = int(cv_threads)
        if replicas is not None:
            hs = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
            _check(lib().mih_cv_iht_multi(hs, len(replicas), C.byref(prm), _p(yv), _p(zz), zz.shape[1], _p(folds), q,
                                          _p(path), path.size, _p(raw)))
        else:
            _check(lib().mih_cv_iht(x._h, C.byref(prm), _p(yv), _p(zz), zz.shape[1], _p(folds), q, _p(path), path.size,
                                    rank, world, _p(raw)))
    if reduce is not None:
        raw = reduce(raw)
    mse = np.zeros(path.size)
    _check(lib().mih_cv_meanloss(_p(np.ascontiguousarray(raw)), _p(folds), n, q, path.size, _p(mse)))
    if verbose and rank == 0:
        print("\n\nCrossvalidation Results:\n\tk\tMSE")
        for kk, m in zip(path, mse):
            print(f"\t{kk}\t{m}")
        print(f"\nBest k = {path[int(np.argmin(mse))]}\n")
    return (mse, raw) if return_raw else mse


# ---- simulation helpers (src/simulate_utilities.jl) ---------------------------------------------------
def simulate_random_snparray(n, p, seed=2024, missing_rate=0.0, device=0):
    """simulate_random_snparray (simulate_utilities.jl:33-47): maf_j ~ U(0, 0.5), g_ij ~ Binomial(2, maf_j), generated
    on the device with a counter-based generator (not Julia's RNG stream); returned ready for fit_iht, i.e. as
    SnpLinAlg{Float64}(x, center=true, scale=true, impute=true)."""
    return SnpLinAlg.synthetic(n, p, seed=seed, missing_rate=missing_rate, center=True, scale=True, impute=True, device=device)


def simulate_random_response(x, k, d, l=None, *, r=10, alpha=1.0, Zu=None, seed=None):
    """simulate_random_response(x, k, d, l; r, α, Zu) -- simulate_utilities.jl:205-244: k effects at random positions
    (N(0,1); N(0,0.3²) for Poisson / Gamma / NegativeBinomial), eta = x*beta + Zu computed on the GPU, y ~ d(linkinv(eta)).
    Returns (y, true_b, correct_position) with 0-based positions."""
    x = _as_mat(x)
    d = _inst(d)
    l = _inst(l) if l is not None else canonicallink(d)
    if isinstance(d, (NegativeBinomial, Gamma)) and not isinstance(l, LogLink):
        raise ArgumentError(f"Distribution {d!r} must use LogLink!")
    rng = np.random.default_rng(seed)
    n, p = x.n, x.p
    small = isinstance(d, (Poisson, Gamma, NegativeBinomial))
    pos = np.sort(rng.choice(p, size=k, replace=False))
    val = rng.normal(0.0, 0.3 if small else 1.0, size=k)
    true_b = np.zeros(p)
    true_b[pos] = val
    eta = x.xv_sparse(pos, val) + (0.0 if Zu is None else np.asarray(Zu, dtype=np.float64).ravel())
    inv = {0: lambda e: e, 1: lambda e: 1 / (1 + np.exp(-e)), 2: np.exp}
    if l.code not in inv:
        raise ArgumentError("simulate_random_response supports the Identity, Logit and Log links")
    mu = np.clip(inv[l.code](eta), -20, 20)
    if isinstance(d, Normal):
        y = mu + rng.standard_normal(n)
    elif isinstance(d, Bernoulli):
        y = (rng.random(n) < mu).astype(np.float64)
    elif isinstance(d, Poisson):
        y = rng.poisson(mu).astype(np.float64)
    elif isinstance(d, NegativeBinomial):
        y = rng.negative_binomial(r, 1.0 / (1.0 + mu / r)).astype(np.float64)
    elif isinstance(d, Gamma):
        y = rng.gamma(alpha, mu / alpha)                 # shape α, rate 1/μ as in the reference
    else:
        raise ArgumentError(f"unsupported distribution {d!r}")
    return y, true_b, pos


def maf_weights(x, max_weight=np.inf):
    """maf_weights(x::SnpArray; max_weight) -- src/utilities.jl:682-697: prior weights 1 / (2 sqrt(p (1 - p)))
    from the minor allele frequencies (SnpArrays.maf: over the non-missing genotypes), clamped to [1, max_weight]."""
    if not isinstance(x, SnpLinAlg):
        raise ArgumentError("maf_weights needs a SnpLinAlg (2-bit genotypes)")
    mu, _ = x.mu_sigma()
    f = mu / 2.0
    maf = np.minimum(f, 1.0 - f)
    with np.errstate(divide="ignore"):
        w = 1.0 / (2.0 * np.sqrt(maf * (1.0 - maf)))
    return np.clip(w, 1.0, max_weight)


def iht_run_many_models(y, x, z=None, *, d=None, l=None, path=range(1, 21), est_r="None", group=None, weight=None,
                        use_maf=False, debias=False, verbose=True, parallel=False, max_iter=100, rank=0, world=1,
                        reduce=None, xtv_digits=None):
    """iht_run_many_models(y, x, z; path, ...) -- src/cross_validation.jl:232-273: fit_iht on the FULL data
    for every model size in `path` (no hold-out), returns the loglikelihoods.  `parallel` (pmap in the
    reference) is accepted and ignored: the fits of one process run back to back on its GPU; `rank` /
    `world` shard `path` over processes and `reduce` sums the loglikelihood vector across them."""
    x = _as_mat(x)
    if _is_multivariate(y):
        raise ArgumentError("iht_run_many_models on the GPU path takes a univariate response")
    d = _inst(d) if d is not None else Normal()
    l = _inst(l) if l is not None else canonicallink(d)          # cross_validation.jl:237
    path = np.ascontiguousarray([int(k) for k in path], dtype=np.int64)
    yv = np.ascontiguousarray(np.asarray(y, dtype=np.float64).ravel())
    n = x.n
    zz = np.ones((n, 1)) if z is None else np.asarray(z, dtype=np.float64)
    zz = np.asfortranarray(zz.reshape(zz.shape[0], -1))
    if not (yv.size == n == zz.shape[0]):
        raise DimensionMismatch(f"row dimension of y, x, and z ({yv.size}, {n}, {zz.shape[0]}) are not equal")
    _checky(yv, d)
    keep = []
    prm = _params(1, 1, d, l, 1e-4, max_iter, 5, 3, est_r, None, weight, group, zz.shape[1], x.p, keep, debias=debias,
                  xtv_digits=xtv_digits)
    logl = np.zeros(path.size)
    _check(lib().mih_fit_iht_path(x._h, C.byref(prm), _p(yv), _p(zz), zz.shape[1], _p(path), path.size, rank, world,
                                  _p(logl), None, None, None))
    if reduce is not None:
        logl = reduce(logl)
    if verbose and rank == 0:
        print("\n\nResults of running many models:\n\tk\tloglikelihood")
        for k, v in zip(path, logl):
            print(f"\t{k}\t{v}")
    return logl


# ---- file-level wrappers (src/wrapper.jl) ----------------------------------------------------
def _phenotype_is_missing(tok):
    return tok in ("-9", "NA")                                   # wrapper.jl:251-253


def _fam_column(prefix, col, d, n):
    """parse_phenotypes(x::SnpData, col, d) -- wrapper.jl:171-214: column `col` (1-based) of the .fam file; missing ("-9", "NA")
    phenotypes are imputed by the mean of the observed ones for quantitative traits and refused for binary / count traits."""
    toks = []
    with open(prefix + ".fam") as f:
        for line in f:
            parts = line.split()
            if parts:
                toks.append(parts[col - 1])
    if len(toks) != n:
        raise DimensionMismatch(f"{prefix}.fam has {len(toks)} samples, expected {n}")
    miss = np.array([_phenotype_is_missing(t) for t in toks])
    if miss.any() and not isinstance(d, (Normal, MvNormal)):
        i = int(np.flatnonzero(miss)[0]) + 1
        raise ArgumentError(f"Missing phenotype detected for sample {i}. Automatic phenotype imputation are only possible for "
                            "quantitative traits. Please exclude missing phenotypes or impute them first.")
    y = np.array([0.0 if m_ else float(t) for t, m_ in zip(toks, miss)])
    if miss.any():
        y[miss] = y[~miss].sum() / (n - miss.sum())
    return y


def _count_lines(path):
    with open(path) as f:
        return sum(1 for line in f if line.strip())


def parse_phenotypes(plinkfile, phenotypes, d, n):
    """parse_phenotypes (wrapper.jl:136-224): .fam column(s) or a comma-separated file, one sample per row."""
    if isinstance(phenotypes, (int, np.integer)):
        if isinstance(d, MvNormal):
            raise ArgumentError("Multivariate analysis requires multiple phenotypes! Please specify e.g. phenotypes=[6, 7] or save each "
                                "sample's phenotypes in a comma-separated file where each sample occupies a different row and each "
                                "phenotype is separated by a single comma.")
        return _fam_column(plinkfile, int(phenotypes), d, n)
    if isinstance(phenotypes, (list, tuple, np.ndarray)):
        return np.stack([_fam_column(plinkfile, int(c), MvNormal(), n) for c in phenotypes])      # r x n
    y = np.loadtxt(phenotypes, delimiter=",", ndmin=2)                                            # readdlm(file, ',', Float64)
    return y.T.copy() if y.shape[1] > 1 else y[:, 0].copy()


def parse_covariates(filename, exclude_std_idx=(), standardize_columns=True):
    """parse_covariates (wrapper.jl:226-249): comma-separated, one sample per row, first column the intercept; every column not in
    exclude_std_idx (1-based indices or a boolean mask) is standardized, the intercept never."""
    z = np.loadtxt(filename, delimiter=",", ndmin=2)
    ex = np.asarray(list(exclude_std_idx))
    std = np.ones(z.shape[1], dtype=bool)
    if ex.dtype == bool:
        std = ~ex
    elif ex.size:
        std[ex.astype(int) - 1] = False
    if np.all(z[:, 0] == 1):
        std[0] = False
    else:
        print("Warning: Covariate file provided but did not detect an intercept. An intercept will NOT be included in IHT!", file=sys.stderr)
    if standardize_columns and std.any():
        z[:, std] = standardize(z[:, std])
    return z


def _read_bim(prefix):
    """chromosome, SNP id, position, allele1, allele2 of the .bim file (what SnpData.snp_info holds, wrapper.jl:451-485)."""
    chrom, ids, pos, a1, a2 = [], [], [], [], []
    with open(prefix + ".bim") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            chrom.append(t[0]); ids.append(t[1]); pos.append(t[3]); a1.append(t[4]); a2.append(t[5])
    return chrom, pos, ids, a1, a2


def _parse_inputs(plinkfile, phenotypes, covariates, d, exclude_std_idx=(), dosage=False, device=0):
    if str(plinkfile).endswith((".vcf", ".vcf.gz", ".bgen")):
        raise ArgumentError("the GPU path reads binary PLINK trios; VCF / BGEN inputs are numeric matrices in the reference "
                            "(wrapper.jl:70-71): convert them to a DenseMatrix and call fit_iht / cv_iht")
    for ext in (".bed", ".bim", ".fam"):
        if not os.path.exists(plinkfile + ext):
            raise ArgumentError(f"{plinkfile}{ext} not found: binary PLINK files should exclude .bim/.bed/.fam trailings and the trio "
                                "must be present in the same directory")
    n = _count_lines(plinkfile + ".fam")
    x = SnpLinAlg(plinkfile + ".bed", n, center=True, scale=True, impute=True, device=device)     # wrapper.jl:68-69
    y = parse_phenotypes(plinkfile, phenotypes, d, n)
    z = parse_covariates(covariates, exclude_std_idx) if covariates else np.ones((n, 1))
    return x, y, z


def _show_result(io, res):
    """show(io, ::IHTResult / ::mIHTResult) -- data_structures.jl:280-325 (the tables are DataFrames in the reference)."""
    if isinstance(res, IHTResult):
        io.write(repr(res) + "\n")
        return
    r = res.traits
    io.write(f"\nCompute time (sec):     {res.time}\nFinal loglikelihood:    {res.logl}\nIterations:             {res.iter}\n")
    for t in range(r):
        io.write(f"Trait {t + 1}'s SNP PVE:      {res.σg[t]}\n")
    io.write("\nEstimated trait covariance:\n")
    io.write("\t".join(f"trait{t + 1}" for t in range(r)) + "\n")
    for row in np.asarray(res.Σ):
        io.write("\t".join(repr(float(v)) for v in row) + "\n")
    for t in range(r):
        for what, arr in (("nonzero SNP predictors", res.beta[t]), ("non-genetic predictors", res.c[t])):
            nz = np.flatnonzero(arr)
            io.write(f"\nTrait {t + 1}: IHT estimated {nz.size} {what}\n Position  Estimated_β\n")
            for j in nz:
                io.write(f" {j + 1:8d}  {arr[j]: .6g}\n")


def iht(plinkfile, k, d, *, phenotypes=6, covariates="", summaryfile="iht.summary.txt", betafile="iht.beta.txt",
        covariancefile="iht.cov.txt", exclude_std_idx=(), dosage=False, device=0, **kwargs):
    """iht(filename, k, d; phenotypes, covariates, summaryfile, betafile, covariancefile, exclude_std_idx, dosage, kwargs...) --
    src/wrapper.jl:52-120 for binary PLINK input: SnpLinAlg{Float64}(center=true, scale=true, impute=true) on the GPU, the
    reference's phenotype / covariate parsing, fit_iht with the canonical link (LogLink for NegativeBinomial, wrapper.jl:87),
    the summary file (the fit's log + show(result)) and the beta file `chr pos SNPid ref alt Estimated_beta` (one row per SNP,
    tab-separated; `beta_1 .. beta_r` columns and the covariance file for multivariate traits, wrapper.jl:100-116).  (v1.4.11
    then overwrites the beta file with an empty CSV header, wrapper.jl:117 `CSV.write(betafile, df)` on an empty DataFrame;
    that accident is not reproduced.)"""
    d = _inst(d)
    x, y, z = _parse_inputs(plinkfile, phenotypes, covariates, d, exclude_std_idx, dosage, device)
    mv = _is_multivariate(y)
    user_io = kwargs.pop("io", None)
    with open(summaryfile, "w") if summaryfile else open(os.devnull, "w") as io:
        if mv:
            result = fit_iht(y, x, z.T, k=k, d=MvNormal(), io=io, **kwargs)
        else:
            l = kwargs.pop("l", None) or (LogLink() if isinstance(d, NegativeBinomial) else canonicallink(d))     # wrapper.jl:87
            result = fit_iht(y, x, z, k=k, d=d, l=l, io=io, **kwargs)
        _show_result(io, result)
    if user_io is not None:
        _show_result(user_io, result)
    if betafile:
        chrom, pos, ids, a1, a2 = _read_bim(plinkfile)
        with open(betafile, "w") as f:
            if mv:
                f.write("chr\tpos\tSNPid\tref\talt" + "".join(f"\tbeta_{t + 1}" for t in range(y.shape[0])) + "\n")
                for j in range(x.p):
                    f.write(f"{chrom[j]}\t{pos[j]}\t{ids[j]}\t{a1[j]}\t{a2[j]}\t" + "\t".join(repr(float(v)) for v in result.beta[:, j]) + "\n")
            else:
                f.write("chr\tpos\tSNPid\tref\talt\tEstimated_beta\n")
                for j in range(x.p):
                    f.write(f"{chrom[j]}\t{pos[j]}\t{ids[j]}\t{a1[j]}\t{a2[j]}\t{float(result.beta[j])!r}\n")
    if covariancefile and mv:
        np.savetxt(covariancefile, np.asarray(result.Σ), delimiter="\t")          # writedlm(covariancefile, result.Σ)
    return result


def print_cv_results(io, errors, path, k):
    """print_cv_results (data_structures.jl:327-335)."""
    io.write("\n\nCrossvalidation Results:\n\tk\tMSE\n")
    for kk, e in zip(path, errors):
        io.write(f"\t{int(kk)}\t{float(e)!r}\n")
    io.write(f"\nBest k = {int(k)}\n\n")


def cross_validate(plinkfile, d, *, path=range(1, 21), q=5, phenotypes=6, covariates="",
                   cv_summaryfile="cviht.summary.txt", exclude_std_idx=(), dosage=False, device=0, **kwargs):
    """cross_validate(filename, d; path, phenotypes, covariates, cv_summaryfile, q, exclude_std_idx, dosage, kwargs...) --
    src/wrapper.jl:301-349 for binary PLINK input; the summary file is print_cv_results + the total time, as the reference's."""
    t0 = time.time()
    d = _inst(d)
    x, y, z = _parse_inputs(plinkfile, phenotypes, covariates, d, exclude_std_idx, dosage, device)
    path = list(path)
    if _is_multivariate(y):
        mse = cv_iht(y, x, z.T, d=MvNormal(), path=path, q=q, **kwargs)
    else:
        l = kwargs.pop("l", None) or (LogLink() if isinstance(d, NegativeBinomial) else canonicallink(d))
        mse = cv_iht(y, x, z, d=d, l=l, path=path, q=q, **kwargs)
    if cv_summaryfile:
        with open(cv_summaryfile, "w") as io:
            print_cv_results(io, mse, path, path[int(np.argmin(mse))])
            io.write(f"Total cross validation time = {time.time() - t0} seconds\n")
    return mse
