#!/usr/bin/env python3
"""bench.py -- IHT iterations/s and X'r GB/s vs the HBM roofline (BASELINE.json metric).

The SAME workload at every N: BASELINE configs[2].  A "step" is one IHT iteration (iht_one_step!, src/fit.jl:213-263) of ONE fit
over a synthetic 2-bit SnpArray resident in HBM (n=500k, p=1M, k=200, Normal): step size (k-column X v), gradient step + top-k
projection, X beta, mean / loglikelihood, backtracking if needed, and the full X'r score pass.  `value` = iterations of that one
fit per second, `"scaling": "strong"`: at N > 1 the SNP columns are split over the ranks (column-sharded fit, SURVEY 8e row 2:
every rank streams p/N columns, the exchanges -- two n-vector sums, one small all-gather, a few scalars per iteration -- run
inside the library over its own RCCL communicator, csrc/comm.hip) and `--steps K` still means K iterations of the one fit.
`roofline` is measured live: HIP events around every launch of the dominant kernel (the X'r pass) inside the timed region, on
the stream it runs on, recorded by the library together with the NAME of the kernel it dispatched (mih_profile_passes); at
N > 1 it is the per-GPU figure (each rank's launches stream its own p/N columns; all ranks' launches are pooled).

Secondary object `cv_iht` at every N: BASELINE configs[3] -- Bernoulli/Logit, path=1:20, 5 folds, all 100 fits -- the path's
natural shard (cross_validation.jl:98-121): every rank holds a full replica of X and evaluates its share of the (fold, k)
combinations, ONE gather of the held-out losses.  At N > 1 rank 0 first times the whole 100-fit cross-validation ALONE, so
the line carries its own N = 1 reference (`cv_iht.one_gpu_s`) and `cv_iht.cv_speedup`; the N-rank losses must equal rank 0's
bit for bit.  `cpu_baseline` (N = 1): the CPU oracle's whole iht_one_step on a bounded column sample of the same matrix;
`cpu_baseline_cv`: the oracle's cv_iht on a reduced grid.  The run FAILS (exit 1) if the fit recovers fewer than 99 % of the
planted effects or the cross-validation does not select the planted model size.

`python bench.py --gpus N` starts its N ranks itself (a child `python -m torch.distributed.run`, before anything touches a GPU)
and relays rank 0's line; started under a launcher (WORLD_SIZE set) it is one of the ranks.  `--mode replicas` keeps round 1's
independent replicas of configs[2] (weak scaling, no exchange).

Prints ONE JSON line (rank 0).  The CPU baselines are a port (the repo's oracle), not MendelIHT.jl: no Julia here.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "IHT iterations/sec + X\u1d40r GB/s vs HBM roofline, n=500k p=1M k=200"       # BASELINE.json's string, verbatim
FP6_PEAK_PFLOPS = 10.0   # dense FP6 / FP4 MFMA peak (MI355X_MICROARCH.md: "~10 PF dense"): the ruler of the fused multi-residual passes
DTYPE = "f64 (residual as a 54-bit fixed-point number, exact accumulation on the matrix cores, f64 recombination)"
TRAFFIC_FILE = os.path.join("profiles", "r06_traffic.json")    # separate rocprofv3 --pmc passes of this command (tools/prof_bench.sh)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed IHT iterations of the one fit (default 100)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=["auto", "fit", "replicas"], default="auto",
                    help="auto = fit: configs[2], one fit, column-sharded over the ranks at N>1; replicas: one independent fit per GPU")
    ap.add_argument("--n", type=int, default=int(os.environ.get("MIH_BENCH_N", 500_000)))
    ap.add_argument("--p", type=int, default=int(os.environ.get("MIH_BENCH_P", 1_000_000)))
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--variant", type=int, default=-1, help="measurement build only (MENDELIHT_HIP_PROBES=1): a round-1 per-wave kernel shape")
    ap.add_argument("--dry-run", action="store_true", help="print each rank's launcher environment and exit (no GPU call)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-started ranks (0 = a free one)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of each baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cv", action="store_true", help="skip the secondary cv_iht measurement (configs[3])")
    ap.add_argument("--no-dense", action="store_true", help="skip the secondary dense-matrix measurement (configs[1], N = 1 only)")
    ap.add_argument("--no-mv", action="store_true", help="skip the secondary multivariate measurement (configs[4], N = 1 only)")
    ap.add_argument("--step-mode", type=int, default=0, choices=[0, 1],
                    help="0: iht_one_step! resident on the device (default), 1: host-driven steps (the path of rounds 1-4)")
    ap.add_argument("--cv-steps", type=int, default=2, help="timed whole cross-validations of the secondary object at N>1")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per X'r launch from a separate rocprofv3 --pmc pass (corrected)")
    return ap.parse_args()


def host_cpus():
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None                                            # cgroup v2 CPU quota of the container, in CPUs
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return ncpu, quota


def cores_used(threads):
    """`cores` of a CPU baseline = the CPUs that actually ran it: min(OpenMP threads, cgroup quota)."""
    _, quota = host_cpus()
    return threads if quota is None else int(min(threads, max(1, round(quota))))


def kfd_gpu_count():
    """GPUs of this node from the kernel driver's topology (no HIP call: the launching process must not open the device).
    None when the topology cannot be read."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir("/sys/class/kfd"):
        return 0                                         # no amdgpu compute driver on this node
    try:
        cnt = 0
        for node in os.listdir(base):
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, node, "properties")) if len(ln.split()) >= 2)
            cnt += int(props.get("simd_count", "0")) > 0          # CPU nodes have no SIMDs
        vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
        if vis is not None and vis.strip() != "":
            cnt = min(cnt, len([v for v in vis.split(",") if v.strip() != ""]))
        return cnt
    except (OSError, ValueError):
        return None


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `python -m torch.distributed.run` (Popen, no
    exec), relay the child's stdout (rank 0's JSON line) and return its exit code.  This process makes no HIP call (ADVICE r3:
    the devices are counted from the kfd topology in sysfs, not with torch.cuda.device_count(), which falls back to
    hipGetDeviceCount and opens the runtime)."""
    if not a.dry_run and not os.environ.get("MIH_BENCH_ONE_DEVICE"):
        have = kfd_gpu_count()
        if have is not None and have < a.gpus:
            sys.exit(f"bench.py --gpus {a.gpus}: this node has {have} GPU(s); one rank per GPU is needed "
                     "(MIH_BENCH_BACKEND=gloo MIH_BENCH_ONE_DEVICE=1 runs all ranks on one device as a functional check)")
    port = a.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")                  # torchrun's default, stated so that it does not warn
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def cpu_baseline(m, n, p, k, seed, target_s):
    """One whole iht_one_step of the oracle (step size, gradient step + projection, X beta, loglikelihood, X'r) on the
    first columns of the same synthetic matrix.  The X'r pass scales with the column count, the rest of the step does
    not: t_step(p) = t_xtv(sample) * p / sample + (t_step(sample) - t_xtv(sample)).  The OpenMP thread count is tuned on a
    probe first (a container CPU quota or a second socket makes more threads slower) and reported as `cores`."""
    from oracle import oracle as O

    ncpu, quota = host_cpus()
    rng = np.random.default_rng(7)
    r = rng.standard_normal(n)

    def load(pc):
        xs = m.SnpLinAlg.synthetic(n, pc, seed=seed)       # same (seed, j) keys => same columns
        cols = xs.export_bed()
        del xs
        return O.Mat.from_bed_columns(cols, n)

    def time_pass(ox, min_wall, max_reps=64):
        ox.xtv(r)                                           # touch / warm
        reps, t0 = 0, time.perf_counter()
        while True:                                         # repeat so the wall-clock sample is not too short
            ox.xtv(r)
            reps += 1
            el = time.perf_counter() - t0
            if el >= min_wall or reps >= max_reps:
                return el / reps

    probe = int(min(p, 4096))
    ox = load(probe)
    trials = {}
    for th in sorted({t for t in (4, 8, 16, 32, 64, 128, 256, ncpu) if t <= ncpu}):
        O.set_threads(th)
        trials[th] = time_pass(ox, 0.2, max_reps=8)
    cores = min(trials, key=trials.get)
    O.set_threads(cores)
    t = trials[cores]
    del ox
    pc = int(min(p, max(probe, probe * (target_s / 4.0) / max(t, 1e-6))))
    #  at least 10 % of the columns when host memory allows: the sample is held twice for a moment (the exported
    # PLINK columns and the oracle's copy)
    avail = 0
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    cap_bytes = max(2e9, min(0.25 * avail, 16e9))
    pc = max(pc, int(min(p // 10, cap_bytes // ((n + 3) // 4))))
    pc = min(pc, max(probe, int(cap_bytes // ((n + 3) // 4))))
    ox = load(pc)
    t_xtv = time_pass(ox, target_s / 4.0)
    # whole iterations: fit_iht with max_iter = M performs M - 1 steps after the initial score pass (fit.jl:170)
    supp = np.sort(rng.choice(pc, size=min(k, pc // 2), replace=False))
    mask = np.zeros(pc, np.uint8)
    mask[supp] = 1
    b = np.zeros(pc)
    b[supp] = rng.standard_normal(supp.size)
    y = ox.xv_masked(mask, b) + 1.0 + rng.standard_normal(n)
    kk = int(supp.size)

    def fit_time(max_iter):
        t0 = time.perf_counter()
        o = O.fit_iht(ox, y, None, k=kk, max_iter=max_iter, tol=1e-15)    # never converges early (the reference requires tol > eps, fit.jl:90)
        return time.perf_counter() - t0, int(o["iter"])
    fit_time(2)
    ta, _ = fit_time(2)
    tb, itb = fit_time(5)
    t_step = (tb - ta) / 3.0                                # three more steps
    t_rest = max(t_step - t_xtv, 0.0)
    t_full = t_xtv * p / pc + t_rest
    return {"value": 1.0 / t_full, "unit": "iterations/s", "cores": cores_used(cores), "omp_threads": cores, "kind": "port",
            "sample": f"one whole oracle iht_one_step (iht_stepsize!, _iht_gradstep!, update_xb!, loglikelihood, score!) with k={kk} "
                      f"on the first {pc} of {p} SNP columns ({100.0 * pc / p:.1f} % of them), n={n}: {t_step:.3f} s per step, of which the X'r pass {t_xtv:.3f} s "
                      f"(scaled by p/{pc}; the remaining {t_rest:.3f} s per step does not grow with p); {cores} OpenMP threads = the "
                      f"fastest of {sorted(trials)} on {ncpu} logical CPUs{'' if quota is None else f', cgroup CPU quota {quota:g}'}; "
                      "CPU restatement (oracle/), not MendelIHT.jl.  Context, not a measurement of this run: the reference's closest "
                      "published point is 2530 s / 4 iterations = 632 s per iteration at n=100k, p=1M, Normal, on one Intel E5-2670 core "
                      "(pre-1.0 code; figures/benchmark/normal_results_nodebias, BASELINE.md)",
            "xtv_GBps": ((n + 3) // 4) * pc / t_xtv / 1e9, "step_s_on_sample": t_step, "xtv_s_on_sample": t_xtv}, cores


def cpu_baseline_cv(m, n, p, seed, cores, gpu_fits_per_s):
    """The oracle's cross-validation fits on a reduced grid of configs[3]: same rows, the first `pc` columns, 5 folds x 3
    model sizes, each fit on its training mask exactly as cv_iht runs it (cross_validation.jl:100-112).  Only the X'r
    passes grow with the column count: full-size time = passes * t_pass(sample) * p / pc + (measured time - passes *
    t_pass(sample)); fits/s at full size follows."""
    from oracle import oracle as O
    hash_folds = m.hash_folds

    O.set_threads(cores)
    pc = 2048
    xs = m.SnpLinAlg.synthetic(n, pc, seed=seed)
    rng = np.random.default_rng(2025)
    supp = np.sort(rng.choice(pc, 10, replace=False))
    eta = xs.xv_sparse(supp, rng.standard_normal(10) * 0.5)
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    ox = O.Mat.from_bed_columns(xs.export_bed(), n)
    del xs
    folds = hash_folds(n, 5)
    path = [5, 10, 15]
    r = rng.standard_normal(n)
    ox.xtv(r)
    t0 = time.perf_counter()
    for _ in range(8):
        ox.xtv(r)
    t_pass = (time.perf_counter() - t0) / 8
    passes, t0 = 0, time.perf_counter()
    for fold in range(1, 6):
        train = (folds != fold).astype(np.uint8)
        for kk in path:
            o = O.fit_iht(ox, yb, None, k=kk, dist="bernoulli", link="logit", max_iter=100, train=train)
            passes += int(o["iter"])                        # one score pass per iteration (the initial one included)
    dt = time.perf_counter() - t0
    fits = 5 * len(path)
    rest = max(dt - passes * t_pass, 0.0)
    full = passes * t_pass * p / pc + rest
    v = fits / full
    ncpu, quota = host_cpus()
    return {"value": v, "unit": "fits/s", "cores": cores_used(cores), "omp_threads": cores, "kind": "port",
            "sample": f"oracle fits of cv_iht Bernoulli/Logit, 5 folds x path={path} ({fits} fits, {passes} X'r passes) on n={n} x the first "
                      f"{pc} of {p} columns: {dt:.2f} s, of which {passes * t_pass:.2f} s in X'r passes ({t_pass * 1e3:.1f} ms each; scaled by "
                      f"p/{pc}) and {rest:.2f} s in work that does not grow with p; {cores} OpenMP threads on {ncpu} logical CPUs"
                      f"{'' if quota is None else f', cgroup CPU quota {quota:g}'}; CPU restatement, not MendelIHT.jl",
            "gpu_over_cpu": (gpu_fits_per_s / v) if v > 0 else None,
            "note": "BASELINE's target is >= 20x over a 2-socket CPU on cv_iht path=1:20; this host exposes a 16-CPU quota of its 2 sockets"}


def cv_problem(m, x, n, p):
    hash_folds = m.hash_folds
    rng = np.random.default_rng(2025)
    supp = np.sort(rng.choice(p, 10, replace=False))
    eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    return yb, hash_folds(n, 5)


CV_PLANTED_K = 10          # cv_problem plants 10 effects; at the default sizes the cross-validation selects exactly that model size


def pass_stats(m, x, passes):
    """Per-launch records of the dominant kernel -> what the roofline object needs: launches, summed and union busy time,
    ALGORITHMIC bytes of exactly those launches (algorithmic_bytes(residuals) each), residual scores, the kernel's name."""
    if not passes:
        return {"launches": 0, "ms_sum": 0.0, "ms_union": 0.0, "bytes": 0.0, "residuals": 0, "kernel": None, "kernels": {}, "digit_columns": 0, "mfma_flop": 0.0}
    by_m = {}
    names = {}
    for q in passes:
        by_m[q["residuals"]] = by_m.get(q["residuals"], 0) + 1
        names[q["kernel"]] = names.get(q["kernel"], 0) + 1
    alg = {mm: x.algorithmic_bytes(mm) for mm in by_m}
    # what the matrix pipe multiplies in a launch: every row of every column against the 32 digit columns of each operand (16 of the
    # last one when the kernel leaves its empty second fragment out: ",half" in the name), 2 flop per multiply-add
    n_pad = (x.n + 127) // 128 * 128
    cols = sum(32 * q["operands"] - (16 if ",half" in q["kernel"] else 0) for q in passes)
    return {"launches": len(passes), "ms_sum": sum(q["ms"] for q in passes), "ms_union": m.busy_union_ms(passes),
            "digit_columns": cols, "mfma_flop": 2.0 * n_pad * x.p * cols,
            "bytes": float(sum(alg[q["residuals"]] for q in passes)), "residuals": int(sum(q["residuals"] for q in passes)),
            "kernel": max(names, key=names.get), "kernels": names}


def cv_roofline(cst):
    return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
            "achieved": cst["bytes"] / (cst["ms_sum"] * 1e-3) / 1e9 if cst["launches"] else None,
            "frac": cst["bytes"] / (cst["ms_sum"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if cst["launches"] else None,
            "kernel": cst["kernel"], "kernel_ms": cst["ms_sum"] / max(cst["launches"], 1),
            "algorithmic_bytes_per_launch": cst["bytes"] / max(cst["launches"], 1),
            "digit_columns_per_launch": cst["digit_columns"] / max(cst["launches"], 1),
            "fp6_PFLOPs": cst["mfma_flop"] / (cst["ms_sum"] * 1e-3) / 1e15 if cst["launches"] else None,
            "frac_of_fp6_peak": cst["mfma_flop"] / (cst["ms_sum"] * 1e-3) / 1e15 / FP6_PEAK_PFLOPS if cst["launches"] else None,
            "note": "sum of algorithmic_bytes(residuals of the launch) / sum of HIP-event durations; the two lock-step lanes' passes "
                    "overlap, so the sum exceeds the union (and may exceed the wall time). A fused pass streams X once for up to 19 "
                    "residuals and is bound by the matrix pipe under the power cap, not by HBM: wider passes LOWER this fraction while "
                    "the cost per residual falls (xtv_kernel_ms_per_residual_scored; DESIGN.md 3.1b).  frac_of_fp6_peak: 2 x n_pad x p x "
                    "(digit columns the launches multiplied) / their summed duration / 10 PF"}


FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X dense FP64 vector peak (MI355X_MICROARCH.md): the ruler for "f64-equivalent" multiply-adds


def mv_object(m, x, n, p, torch, r=10, k=500, comm=None, lo=0, sum_over_ranks=None):
    """BASELINE configs[4]: multivariate IHT (MvNormal, r = 10 traits, k = 500) on the resident matrix -- the 10-residual fused
    X'(Y - mu) pass and a whole iteration, timed inside a short fit (the library's HIP events around every pass).
    comm: x is this rank's block of the p columns (first column lo) and the fit is column-sharded (round 5: mih_fit_mv takes a
    shard); the planted X B is summed over the ranks once, at set-up."""
    rng = np.random.default_rng(3)
    lin = rng.choice(r * p, k, replace=False)
    Y = rng.standard_normal((r, n))
    for t in range(r):
        cols = np.unique(lin[lin % r == t] // r)
        eff = rng.choice([-1.0, 1.0], cols.size) * rng.uniform(0.15, 0.45, cols.size)
        mine = (cols >= lo) & (cols < lo + x.p)
        xb = x.xv_sparse(cols[mine] - lo, eff[mine])
        Y[t] += (sum_over_ranks(xb) if comm is not None else xb) + 1.0
    fit = (lambda **kw: m.fit_iht(Y, x, None, k=k, verbose=False, comm=comm, **kw)) if comm is not None else \
          (lambda **kw: m.fit_iht(Y, x, None, k=k, verbose=False, **kw))
    fit(max_iter=3)                                                        # warm-up
    m.profile_read(x, reset=True)
    m.profile_enable(x, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fit(max_iter=12)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    m.profile_enable(x, False)
    ps = m.profile_passes(x, reset=True)                                   # [0] = the initial score
    steady = ps[1:]
    pass_ms = sum(q["ms"] for q in steady) / max(len(steady), 1)
    gaps = [ps[i + 1]["start_ms"] - ps[i]["start_ms"] - ps[i]["ms"] for i in range(1, len(ps) - 1)]
    alg = x.algorithmic_bytes(r)
    n_pad = (n + 127) // 128 * 128
    cols = sum(32 * q["operands"] - (16 if ",half" in q["kernel"] else 0) for q in steady) / max(len(steady), 1)
    outside = sum(gaps) / max(len(gaps), 1)
    auto = None
    if comm is None:            # the same fit with xtv_digits = -1: per pass, the 43-bit format when every trait's T1 row passes the guard
        fit(max_iter=3, xtv_digits=-1)
        m.profile_read(x, reset=True)
        m.profile_counters(x, reset=True)
        m.profile_enable(x, True)
        torch.cuda.synchronize()
        res2 = fit(max_iter=12, xtv_digits=-1)
        torch.cuda.synchronize()
        m.profile_enable(x, False)
        ps2 = m.profile_passes(x, reset=True)
        c2 = m.profile_counters(x, reset=True)
        st2 = ps2[1:]
        nz = res.beta != 0
        auto = {"what": "mih_fit_params::xtv_digits = -1: per pass, the 43-bit fixed-point format (four residuals per operand) when every "
                        "trait's row of T1 = Gamma * resid has max|t| <= 128 rms(t), the 54-bit format otherwise",
                "iterations": int(res2.iter), "ms_per_iteration": 1e3 * res2.time / max(res2.iter, 1),
                "pass_ms": sum(q["ms"] for q in st2) / max(len(st2), 1), "pass_kernel": st2[0]["kernel"] if st2 else None,
                "residuals_in_the_43_bit_format": int(c2["residuals_43bit"]), "passes": len(ps2),
                "same_support": bool(np.array_equal(res2.beta != 0, nz)),
                "largest_relative_difference_of_an_effect": float(np.max(np.abs(res2.beta[nz] - res.beta[nz]) / np.abs(res.beta[nz]))) if nz.any() else 0.0}
    return {"auto_digits": auto, "workload": f"fit_iht MvNormal r={r} traits k={k} on the same SnpArray n={n} p={p} (BASELINE configs[4]), max_iter=12"
                        + (f"; SNP columns sharded over {comm.world} ranks (this rank: {x.p} columns; per-rank pass and bytes)" if comm is not None else ""),
            "iterations": int(res.iter), "ms_per_iteration": 1e3 * res.time / max(res.iter, 1), "fit_wall_s": wall,
            "pass_ms": pass_ms, "outside_the_pass_ms": outside, "ms_per_step_with_pass": pass_ms + outside,
            "ms_per_iteration_note": "ms_per_iteration is the reference's .time / .iter: a fit of N iterations runs N - 1 step passes (the converging "
                                     "step's score is skipped), so it is SMALLER than a step; ms_per_step_with_pass = pass + what lies between two passes",
            "pass_kernel": steady[0]["kernel"] if steady else None,
            "nonzero": int(np.count_nonzero(res.beta)),
            "roofline": {"bound": "mfma (matrix pipe under the package power cap; DESIGN.md 3.1b)", "hbm_GBps": alg / pass_ms / 1e6,
                         "frac_of_hbm_peak": alg / pass_ms / 1e6 / HBM_PEAK_GBPS,
                         "f64_equivalent_TFLOPs": 2.0 * n * x.p * r / (pass_ms * 1e-3) / 1e12,
                         "frac_of_fp64_vector_peak": 2.0 * n * x.p * r / (pass_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "digit_columns_per_launch": cols, "fp6_PFLOPs": 2.0 * n_pad * x.p * cols / (pass_ms * 1e-3) / 1e15,
                         "frac_of_fp6_peak": 2.0 * n_pad * x.p * cols / (pass_ms * 1e-3) / 1e15 / FP6_PEAK_PFLOPS,
                         "note": "2 n p r multiply-adds per pass, each exact (fixed-point residual digits on the matrix cores): what a "
                                 "dense f64 X'R would need 78.6 TFLOP/s of vector FMAs for"}}


def dense_object(m, torch, n=50_000, p=100_000):
    """BASELINE configs[1]: the X'r pass over a dense Matrix{Float64} (k_xtv_dense_lds<f64>), 40 GB streamed once per pass."""
    xd = m.DenseMatrix.synthetic(n, p, seed=2024)
    torch.cuda.synchronize()
    ms, _cs = xd.bench_xtv(iters=20, warmup=3)
    alg = xd.algorithmic_bytes(1)
    res = m.fit_iht(xd.xv_sparse(np.arange(0, p, p // 100)[:100], np.linspace(-1.0, 1.0, 100)) + np.random.default_rng(1).standard_normal(n),
                    xd, None, k=100, verbose=False, max_iter=12)
    del xd
    return {"workload": f"fit_iht Float64 dense x randn({n},{p}) k=100 Normal (BASELINE configs[1]): the X'r pass, HIP events over 20 launches",
            "pass_ms": ms, "algorithmic_bytes_per_launch": alg,
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "achieved": alg / (ms * 1e-3) / 1e9,
                         "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel": "k_xtv_dense_lds<f64>"},
            "fit": {"iterations": int(res.iter), "ms_per_iteration": 1e3 * res.time / max(res.iter, 1), "nonzero": int(np.count_nonzero(res.beta))}}


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("--gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not launched:
        #  the plain command is the one the driver runs: start the ranks here, before any GPU call
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        # an inconsistent launcher environment must not measure something else silently
        sys.exit(f"bench.py --gpus {a.gpus} was started by a launcher with WORLD_SIZE={world}: start it as\n"
                 f"  python bench.py --gpus {a.gpus} [--steps K --warmup W]      (it starts its own ranks), or\n"
                 f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 "
                 f"--master-port 29500 bench.py --gpus {a.gpus} [--steps K --warmup W]")
    if a.dry_run:
        sys.stdout.write(json.dumps({"dry_run": True, "rank": rank, "local_rank": local, "world_size": world,
                                     "master_addr": os.environ.get("MASTER_ADDR"), "master_port": os.environ.get("MASTER_PORT"),
                                     "n_gpus": a.gpus, "workload": "configs[2], one fit" + (", column-sharded" if world > 1 else "")}) + "\n")
        sys.stdout.flush()      # one write per rank: the ranks share the launcher's pipe
        return
    mode = "fit" if a.mode == "auto" else a.mode
    if a.steps is None:
        a.steps = 100
    if a.warmup is None:
        a.warmup = 5
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # MIH_BENCH_BACKEND=gloo + MIH_BENCH_ONE_DEVICE=1: multi-rank smoke test on a single-GPU box
    backend = os.environ.get("MIH_BENCH_BACKEND", "nccl")
    one_device = bool(os.environ.get("MIH_BENCH_ONE_DEVICE"))
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    import mendeliht_amd as m
    from mendeliht_amd import dist as D
    if not os.path.exists(m.library_path()):       # fresh checkout: compile the HIP library first
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if world > 1:
            dist.barrier()
    tdev = "cuda" if backend == "nccl" and world > 1 else "cpu"

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if world == 1:
            return arr
        t = torch.from_numpy(arr.copy()).to(tdev)
        dist.all_reduce(t)
        return t.cpu().numpy()

    def gather_rows(row):
        """every rank's list of floats, as a list of lists on every rank"""
        if world == 1:
            return [list(row)]
        t = torch.tensor(row, dtype=torch.float64, device=tdev)
        allr = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        return [q.cpu().tolist() for q in allr]

    n, p, k = a.n, a.p, a.k
    if a.variant >= 0:
        m.probe_set(variant=a.variant)                      # raises unless the measurement build is loaded
    sharded = mode == "fit" and world > 1
    seed = 2024 + (rank if mode == "replicas" else 0)
    lo, cnt = D.column_block(p, rank, world) if sharded else (0, p)
    t_gen = time.perf_counter()
    x = m.SnpLinAlg.synthetic(n, cnt, seed=seed, device=local, col_offset=lo)     # this rank's block of the SAME matrix (keyed by global column)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    # ---- BASELINE configs[2]: one fit, step = iht_one_step! --------------------------------------------------------------
    # phenotype: y = X beta + 1 + N(0,1), k true effects ~ N(0,1)  (simulate_utilities.jl:215-228); replicated on every rank
    rng = np.random.default_rng(2025 + (rank if mode == "replicas" else 0))
    supp = np.sort(rng.choice(p, size=k, replace=False))
    beta = rng.standard_normal(k)
    mine = (supp >= lo) & (supp < lo + cnt)
    xb = x.xv_sparse(supp[mine] - lo, beta[mine])
    if sharded:
        xb = sum_over_ranks(xb)                             # set-up, untimed: the planted X beta over all shards
    y = xb + 1.0 + rng.standard_normal(n)

    comm = None
    exchange = None
    if sharded:
        # the library's own RCCL communicator (one GPU per rank); on the one-GPU smoke box the torch.distributed callbacks over
        # gloo -- or, with MENDELIHT_RCCL_LIB pointing at tests/libfake_rccl.so, the same native code over the stand-in
        native = backend == "nccl" or bool(os.environ.get("MENDELIHT_RCCL_LIB"))
        native_error = None
        if native:
            try:
                comm = D.NativeComm(lo, p, device=local)
            except Exception as e:      # noqa: BLE001 -- e.g. librccl not loadable: every rank falls back together, and the line says so
                native_error = repr(e)
            if sum_over_ranks([0.0 if native_error is None else 1.0])[0] > 0:
                if comm is not None:
                    comm.close()
                native, comm = False, None
        if not native:
            comm = D.ColumnComm(lo, p, device=local)
        exchange = ("native RCCL inside the library (mih_comm_create_rccl: ncclAllReduce / ncclAllGather on a private stream)" if native
                    else f"torch.distributed callbacks ({backend})"
                         + (f" -- the library's own communicator could not be created: {native_error}" if native_error else ""))
    # same-box A/B of the two ways a step is driven (N = 1): a short run of host-driven steps on a fresh session
    # (round 6: BEFORE the timed session exists -- beside a resident session that has just run, and later in the process when the chip
    # is warmer, the same steps measured 0.46-0.69 ms outside the pass instead of 0.30-0.33: tools/coexist_probe.py)
    ab = None
    if world == 1 and a.step_mode == 0:
        s2 = m.IHTSession(y, x, None, k=k, d=m.Normal(), l=m.IdentityLink(), step_mode=1)
        for _ in range(a.warmup):
            s2.step()
        nab = min(a.steps, 20)
        m.profile_read(x, reset=True)
        m.profile_enable(x, True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _l2, nbt2, _t2 = s2.run(nab)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        m.profile_enable(x, False)
        st2 = pass_stats(m, x, m.profile_passes(x, reset=True))
        s2.close()
        del s2
        k2 = st2["ms_sum"] / max(st2["launches"], 1)
        ab = {"steps": nab, "ms_per_step": 1e3 * el2 / nab, "xtv_kernel_ms": k2, "outside_the_pass_ms_per_step": 1e3 * el2 / nab - k2,
              "backtracks": int(nbt2), "what": "mih_fit_params::step_mode = 1: the host-driven step of rounds 1-4 (26 launches, three host "
              "waits per step without backtracking), same box, same fit, a fresh session"}

    sess = m.IHTSession(y, x, None, k=k, d=m.Normal(), l=m.IdentityLink(), comm=comm, step_mode=a.step_mode)
    for _ in range(a.warmup):
        sess.step()
    m.profile_read(x, reset=True)
    m.profile_counters(x, reset=True)
    m.profile_exchange(x, reset=True)
    m.profile_enable(x, True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    logl, nbt, tol = sess.run(a.steps)          # EXACTLY a.steps iterations, looped inside the library (mih_session_run)
    torch.cuda.synchronize()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    m.profile_enable(x, False)
    st = pass_stats(m, x, m.profile_passes(x, reset=True))
    step_counters = m.profile_counters(x, reset=True)
    exch = m.profile_exchange(x, reset=True)
    comm_info = comm.info() if (comm is not None and hasattr(comm, "info")) else None

    # ... and of what the measurement itself costs (N = 1): the same session goes on for a few steps with the hook OFF.  The hook
    # brackets every X'r pass with two HIP event records, queue operations of their own inside the step chain; the timed region
    # above carries them (the roofline is measured there), this figure says what a step takes without them
    unhooked = None
    if world == 1:
        nun = min(a.steps, 20)
        # the pass kernel's duration beside these steps: 20 hooked steps first (the chip is warmer than in the timed region, and the
        # kernel slower by 0.1-0.2 ms: subtracting the timed region's figure charged that to the chain)
        m.profile_read(x, reset=True)
        m.profile_enable(x, True)
        sess.run(nun)
        m.profile_enable(x, False)
        st3 = pass_stats(m, x, m.profile_passes(x, reset=True))
        kern_ms_here = st3["ms_sum"] / max(st3["launches"], 1)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        logl_after, _nbt3, _tol3 = sess.run(nun)
        torch.cuda.synchronize()
        el3 = time.perf_counter() - t2
        unhooked = {"steps": nun, "logl_after_these_steps": logl_after, "ms_per_step": 1e3 * el3 / nun, "xtv_kernel_ms_of_the_20_hooked_steps_before": kern_ms_here,
                    "outside_the_pass_ms_per_step": 1e3 * el3 / nun - kern_ms_here,
                    "what": "the same fit continued with the measurement hook off (no HIP event records around the passes); the pass kernel's "
                            "duration is that of 20 hooked steps run right before (the same temperature)"}
    #  the model is read AFTER the continuation (reading it brings the iterate home and the chain would have to warm up again):
    # true_effects_recovered describes the iterate whose loglikelihood is without_the_measurement_hook.logl_after_these_steps
    bhat, _ = sess.model()
    found = float(np.intersect1d(np.flatnonzero(bhat) + lo, supp).size)
    recovered = int(sum_over_ranks([found])[0]) if sharded else int(found)

    sess.close()
    del sess                                                # (it holds a reference to the matrix: the shard must be gone before the full replica comes)
    # ---- secondary at N > 1: BASELINE configs[4] column-sharded over the same ranks, through the same communicator ----------
    mv_sharded = None
    if sharded and comm is not None and not a.no_mv:
        try:
            mv_sharded = mv_object(m, x, n, p, torch, comm=comm, lo=lo, sum_over_ranks=sum_over_ranks)
        except Exception as e:      # noqa: BLE001 -- reported in the line, the headline stands
            mv_sharded = {"error": repr(e)}
    if comm is not None and hasattr(comm, "close"):
        comm.close()                                        # collectively, while every rank is alive
    rows = gather_rows([st["launches"], st["ms_sum"], st["bytes"], float(cnt)]) if sharded else [[st["launches"], st["ms_sum"], st["bytes"], float(cnt)]]
    kernel_name = st["kernel"]
    failures = []

    out = None
    if rank == 0:
        traffic_src = None
        if a.traffic_bytes is None and world == 1:      # PMC traffic comes from a separate rocprofv3 pass of this command (profiles/)
            try:
                t = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
                if t["workload"] == {"n": n, "p": p} and t["kernel"] == st["kernel"]:
                    a.traffic_bytes = t["hbm_bytes_per_launch"]
                    traffic_src = (f"{TRAFFIC_FILE} (separate rocprofv3 --pmc passes of this command, kernel {t['kernel']}; "
                                   "not collected by this run)")
            except (OSError, KeyError, ValueError):
                pass
        elif a.traffic_bytes is not None:
            traffic_src = "--traffic-bytes"
        launches = sum(r[0] for r in rows)
        ms_sum = sum(r[1] for r in rows)
        nbytes = sum(r[2] for r in rows)
        alg_bytes = nbytes / launches if launches else x.algorithmic_bytes(1)
        kern_ms = ms_sum / max(launches, 1)                # per launch of ONE rank (each rank streams its own columns)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if launches else 0.0
        nfit = world if mode == "replicas" else 1
        out = {
            "metric": METRIC,
            "value": nfit * a.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak" if mode == "replicas" else "strong",
            "vs_baseline": None,
            "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": f"iht on synthetic SnpArray n={n} p={p} k={k} Normal/Identity (BASELINE configs[2])"
                                   + (f": ONE fit, SNP columns sharded over {world} GPUs" if sharded else "")
                                   + (", one independent replica per GPU" if mode == "replicas" and world > 1 else ""),
                       "n": n, "p": p, "k": k, "xtv_variant": a.variant, "generator_s": round(t_gen, 2),
                       "backtracks_in_timed_steps": nbt, "true_effects_recovered": f"{recovered}/{k}",
                       "final_logl": logl, "host_small_kernels_and_exchange_ms_per_step": 1e3 * elapsed / a.steps - kern_ms,
                       "step_mode": ("0: iht_one_step! resident on the device -- iterate, top-k finish, backtracking decision in device "
                                     "memory, the kernels of a step queued without a host wait" if a.step_mode == 0 and not sharded else
                                     ("0: iht_one_step! resident on the device, the shards' all-reduces and the projection's all-gather queued inside "
                                      "the gated chain (the library's own communicator)" if sharded and a.step_mode == 0 and step_counters["resident_steps"] > 0
                                      else "1: host-driven steps" + (" (callbacks of the host language need the host)" if sharded and a.step_mode == 0 else ""))),
                       "resident_steps": {kk: step_counters[kk] for kk in ("resident_steps", "resident_attempts", "resident_handbacks",
                                                                            "resident_direct", "resident_redos")}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": a.traffic_bytes, "traffic_source": traffic_src,
                         "kernel": kernel_name, "kernel_ms": kern_ms, "launches": int(launches),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if ab is not None:
            out["config"]["host_driven_steps_same_box"] = ab
        if unhooked is not None:
            out["config"]["without_the_measurement_hook"] = unhooked
        # what is outside the pass, split: the exchanges (per kind, timed on the fit's stream / the host clock by the library's hook)
        # and the chain of small kernels
        ex_ms = sum(v["ms"] for v in exch.values()) / a.steps
        out["config"]["exchange_ms_per_step"] = ex_ms
        out["config"]["chain_ms_per_step"] = out["config"]["host_small_kernels_and_exchange_ms_per_step"] - ex_ms
        if sharded:
            out["config"]["collectives_rank0"] = {kk: {"per_step": v["count"] / a.steps, "mean_us": 1e3 * v["ms"] / max(v["count"], 1)}
                                                  for kk, v in exch.items()}
            out["config"]["rccl_ranks_seen"] = comm_info[0] if comm_info else None
            out["config"]["librccl"] = comm_info[1] if comm_info else None
        if sharded:
            out["config"].update(columns_per_rank=[int(r[3]) for r in rows], exchange=exchange,
                                 exchanges_per_iteration="two all-reduces of an n-vector (X_S b_S of update_xb!, X_S g_S of iht_stepsize!), one "
                                                         "all-gather of k top-k candidates per rank (project_k!), a few scalars")
            out["roofline"]["note"] = (f"per GPU: every rank's launches stream its own {cnt} of {p} columns; algorithmic bytes and HIP-event "
                                       "durations pooled over the launches of all ranks")
            out["per_rank"] = [{"rank": i, "columns": int(r[3]), "xtv_launches": int(r[0]), "xtv_kernel_ms": r[1] / max(r[0], 1),
                                "xtv_GBps": r[2] / max(r[1], 1e-9) / 1e6} for i, r in enumerate(rows)]
        if mode != "replicas" and recovered < 0.99 * k:
            failures.append(f"the fit recovered {recovered} of {k} planted effects (< 99 %)")

    # ---- secondary: BASELINE configs[3] on full replicas of X (the path's natural shard), with its own one-GPU reference ----
    gpu_fits_per_s = None
    run_cv = mode == "fit" and not a.no_cv
    if run_cv and one_device and world > 1:
        _free, tot = torch.cuda.mem_get_info()
        if world * (p * ((n + 3) // 4)) * 1.3 > tot:             # the ranks share ONE device: full replicas may not fit beside each other
            run_cv = False
            if rank == 0:
                out["cv_iht"] = {"skipped": f"{world} full replicas of X do not fit the one shared device of this smoke run"}
    if run_cv:
        if sharded:
            del x
            import gc
            gc.collect()
            x = m.SnpLinAlg.synthetic(n, p, seed=2024, device=local)          # every rank: the whole matrix
            torch.cuda.synchronize()
        yb, folds = cv_problem(m, x, n, p)
        path = range(1, 21)

        def one_cv(rk, wd, reduce):
            t0 = time.perf_counter()
            mse, raw = m.cv_iht(yb, x, None, path=path, q=5, folds=folds, verbose=False, return_raw=True, rank=rk, world=wd,
                                reduce=reduce, d=m.Bernoulli(), l=m.LogitLink())
            return mse, raw, time.perf_counter() - t0

        # (1) the whole cross-validation on ONE GPU: the N = 1 line's object, and the in-run reference of an N > 1 line
        solo = None
        if rank == 0:
            one_cv(0, 1, None)                                  # warm-up: first-call work (worker streams, workspaces out of the reserve)
            m.profile_read(x, reset=True)
            m.profile_counters(x, reset=True)
            m.profile_enable(x, True)
            mse1, raw1, dt1 = one_cv(0, 1, None)
            m.profile_enable(x, False)
            cst = pass_stats(m, x, m.profile_passes(x, reset=True))
            cnt1 = m.profile_counters(x, reset=True)
            gpu_fits_per_s = 100.0 / dt1
            best_k = int(np.argmin(mse1)) + 1
            solo = {"workload": "cv_iht Bernoulli/Logit path=1:20, 5 folds (BASELINE configs[3]), all 100 fits on ONE GPU",
                    "seconds": dt1, "cv_iht_s": dt1, "warmup_runs": 1, "fits": int(np.count_nonzero(raw1)), "fits_per_s": gpu_fits_per_s,
                    "iterations": cnt1["scores"], "iterations_per_s": cnt1["scores"] / dt1,
                    "initial_scores": cnt1["init_scores"], "best_k": best_k,
                    "fused_passes": cst["launches"], "residuals_scored_by_passes": cst["residuals"],
                    "xtv_busy_union_ms": cst["ms_union"], "xtv_kernel_ms_sum_over_lanes": cst["ms_sum"],
                    "xtv_kernel_ms_per_residual_scored": cst["ms_sum"] / max(cst["residuals"], 1),
                    "lockstep": cnt1, "kernels": cst["kernels"], "roofline": cv_roofline(cst)}
            if (n, p) == (500_000, 1_000_000) and best_k != CV_PLANTED_K:
                failures.append(f"cv_iht selected k = {best_k}, the planted model has {CV_PLANTED_K} effects")
            solo["dtype"] = DTYPE
            # the same cross-validation with xtv_digits = -1: every residual whose max |r| <= 128 rms(r) is scored in the 43-bit format
            # (all of them here: |y - mu| < 1), four per operand instead of three
            t0 = time.perf_counter()
            m.cv_iht(yb, x, None, path=path, q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink(), xtv_digits=-1)
            m.profile_counters(x, reset=True)
            m.profile_read(x, reset=True)
            m.profile_enable(x, True)
            t0 = time.perf_counter()
            mseA, rawA = m.cv_iht(yb, x, None, path=path, q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink(),
                                  xtv_digits=-1)
            dtA = time.perf_counter() - t0
            m.profile_enable(x, False)
            cstA = pass_stats(m, x, m.profile_passes(x, reset=True))
            cntA = m.profile_counters(x, reset=True)
            solo["auto_digits"] = {
                "what": "mih_fit_params::xtv_digits = -1: per residual, the 43-bit fixed-point format (8 base-49 FP6 digits, four residuals per "
                        "operand) when max|r| <= 128 rms(r), the 54-bit format otherwise; warm-up + 1 run",
                "dtype": "f64 results from residuals rounded to 43 bits of max|r| where the residual's range allows it (else 54 bits); exact "
                         "accumulation on the matrix cores, f64 recombination",
                "cv_iht_s": dtA, "fits_per_s": 100.0 / dtA, "best_k": int(np.argmin(mseA)) + 1, "fused_passes": cstA["launches"],
                "residuals_scored_by_passes": cstA["residuals"], "residuals_in_the_43_bit_format": cntA["residuals_43bit"],
                "xtv_kernel_ms_per_residual_scored": cstA["ms_sum"] / max(cstA["residuals"], 1), "kernels": cstA["kernels"],
                "largest_relative_difference_of_a_loss_to_the_54_bit_run": float(np.max(np.abs(rawA - raw1) / np.abs(raw1)))}
        if world == 1:
            out["cv_iht"] = solo
        else:
            # (2) the same cross-validation over the N ranks: (fold, k) combinations dealt out by mih_cv_assignment, one gather
            native_cv = native if sharded else (backend == "nccl" or bool(os.environ.get("MENDELIHT_RCCL_LIB")))
            gcomm = D.NativeComm(0, 1, device=local) if native_cv else None
            reduce = D.gather_losses_native(gcomm) if native_cv else D.gather_losses
            barrier()
            one_cv(rank, world, reduce)                         # warm-up (first-call workspaces on the other ranks)
            m.profile_read(x, reset=True)
            m.profile_counters(x, reset=True)
            m.profile_enable(x, True)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(1, a.cv_steps)):
                mseN, rawN, _dt = one_cv(rank, world, reduce)
            torch.cuda.synchronize()
            barrier()
            dtN = max_over_ranks(time.perf_counter() - t0) / max(1, a.cv_steps)
            m.profile_enable(x, False)
            cstN = pass_stats(m, x, m.profile_passes(x, reset=True))
            cntN = m.profile_counters(x, reset=True)
            if gcomm is not None:
                gcomm.close()
            rowsN = gather_rows([cstN["launches"], cstN["ms_sum"], cstN["bytes"], cstN["residuals"], cntN["scores"], cntN["fits"]])
            if rank == 0:
                K = max(1, a.cv_steps)
                same = bool(np.array_equal(rawN, raw1))
                out["cv_iht"] = {
                    "workload": f"cv_iht Bernoulli/Logit path=1:20, 5 folds (BASELINE configs[3]): 100 (fold,k) fits sharded over {world} ranks "
                                "(mih_cv_assignment), identical X replica per GPU, ONE gather of the held-out losses per run inside the "
                                "timed region (cross_validation.jl:98-121)",
                    "scaling": "strong", "cv_iht_s": dtN, "one_gpu_s": solo["cv_iht_s"], "cv_speedup": solo["cv_iht_s"] / dtN,
                    "fits_per_s": 100.0 / dtN, "iterations_per_s": sum(r[4] for r in rowsN) / K / dtN,
                    "best_k": int(np.argmin(mseN)) + 1, "losses_equal_one_gpu_run_bit_for_bit": same,
                    "gather": "mih_cv_allgather (ncclAllGather inside the library)" if native_cv else f"torch.distributed all-gather ({backend})",
                    "per_rank": [{"rank": i, "fits": r[5] / K, "iterations": r[4] / K, "fused_passes": r[0] / K,
                                  "xtv_kernel_ms": r[1] / K} for i, r in enumerate(rowsN)],
                    "one_gpu": solo,
                }
                if not same:
                    failures.append("the losses of the sharded cross-validation differ from the one-GPU run")
    # ---- secondary: BASELINE configs[4] (N = 1): MvNormal, r = 10 traits, k = 500 on the same matrix ------------------------
    if world == 1 and mode == "fit" and not a.no_mv and (n, p) == (500_000, 1_000_000):
        out["mv"] = mv_object(m, x, n, p, torch)
    elif rank == 0 and mv_sharded is not None:
        out["mv"] = mv_sharded
    # ---- secondary: BASELINE configs[1] (N = 1): the dense Matrix{Float64} pass, randn(50000, 100000) --------------------------
    if world == 1 and mode == "fit" and not a.no_dense and (n, p) == (500_000, 1_000_000):
        try:
            out["dense"] = dense_object(m, torch)
        except Exception as e:      # noqa: BLE001 -- reported in the line, the headline stands
            out["dense"] = {"error": repr(e)}
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], cores = cpu_baseline(m, n, p, k, seed, a.cpu_seconds)
            if gpu_fits_per_s is not None:
                out["cpu_baseline_cv"] = cpu_baseline_cv(m, n, p, seed, cores, gpu_fits_per_s)
        if failures:
            out["failed"] = failures
        print(json.dumps(out), flush=True)
    nfail = int(sum_over_ranks([float(len(failures))])[0]) if world > 1 else len(failures)
    del x
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if nfail:
        sys.exit(1)


if __name__ == "__main__":
    main()
