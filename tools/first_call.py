"""The first cv_iht of a fresh process against the second, at BASELINE configs[3] size (MIH_P columns, default 200k; the incident
needs MIH_P=1000000).  Before the matrix kept a reserve of device memory for its fits (DevPool; MENDELIHT_NO_RESERVE=1 switches it
off), 7 of 16 fresh processes lost ~2.9 s inside one hipMalloc of the first call: the driver clearing a large never-used block of
VRAM.  MIH_SLEEP=s pauses before the first call, MIH_PRIME=1 allocates and releases 16 GB first (the experiments that located it);
MENDELIHT_CV_TRACE=1 prints the lock-step rounds.  Run it in a shell loop and compare FIRST with SECOND."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
hash_folds = m.hash_folds
n, p = 500_000, int(os.environ.get("MIH_P", 200_000))
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = hash_folds(n, 5)
if os.environ.get("MIH_SLEEP"):
    time.sleep(float(os.environ["MIH_SLEEP"]))
if os.environ.get("MIH_PRIME"):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    tp = time.perf_counter()
    ptrs = []
    for nbytes in [2 << 30] * 4 + [128 << 20] * 64:
        ptr = ctypes.c_void_p()
        if hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(nbytes)) == 0: ptrs.append(ptr)
    for ptr in ptrs: hip.hipFree(ptr)
    print(f"PRIME pattern {len(ptrs)} blocks in {time.perf_counter() - tp:.3f} s", file=sys.stderr, flush=True)
print(f"PY before first call at {time.monotonic():.3f}", file=sys.stderr, flush=True)
t0 = time.perf_counter()
m.cv_iht(yb, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, d=m.Bernoulli(), l=m.LogitLink())
t1 = time.perf_counter()
print(f"PY after first call at {time.monotonic():.3f}", file=sys.stderr, flush=True)
m.cv_iht(yb, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, d=m.Bernoulli(), l=m.LogitLink())
t2 = time.perf_counter()
print(f"FIRST {t1 - t0:.3f} s SECOND {t2 - t1:.3f} s", file=sys.stderr, flush=True)
