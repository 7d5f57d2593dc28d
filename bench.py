#!/usr/bin/env python3
"""bench.py -- IHT iterations/s and X'r GB/s vs the HBM roofline (BASELINE.json metric).

N = 1 (default): BASELINE configs[2].  A "step" is one IHT iteration (iht_one_step!, src/fit.jl:213-263) over a
synthetic 2-bit SnpArray resident in HBM (n=500k, p=1M, k=200, Normal): step size (k-column X v), gradient step +
top-k projection, X beta, mean / loglikelihood, backtracking if needed, and the full X'r score pass.  `roofline` is
measured live: HIP events around every launch of the dominant kernel (the X'r pass) inside the timed region, on the
stream it runs on.  Secondary objects in the same line: `cv_iht` (BASELINE configs[3] -- Bernoulli/Logit, path=1:20,
5 folds, all 100 fits -- on this one GPU, same matrix), `cpu_baseline` (the CPU oracle's whole iht_one_step on a bounded
column sample of the same matrix) and `cpu_baseline_cv` (the oracle's cv_iht on a reduced grid of configs[3]).

N > 1 (launched by torch.distributed.run, one rank per GPU): the path's real shard, BASELINE configs[3] --
cross_validation.jl:98-121.  Every rank holds an identical replica of X; a step is ONE whole cv_iht (100 (fold,k) fits)
strong-scaled over the ranks (combination i -> rank i mod N) with the single RCCL all-gather of the held-out losses
inside the timed region.  `value` = IHT iterations (summed over the fits of all ranks) per second; fits/s, passes per
rank and the gather time are reported beside it.  `--mode replicas` keeps round 1's independent replicas of configs[2].

Prints ONE JSON line (rank 0).  The CPU baselines are a port (the repo's oracle), not MendelIHT.jl: no Julia here.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "IHT iterations/sec + X'r GB/s vs HBM roofline, n=500k p=1M k=200"
DTYPE = "f64 (residual as a 54-bit fixed-point number, exact accumulation on the matrix cores, f64 recombination)"
XTV_KERNEL = "k_xtv_dma<1,2,4,8,false,0>"      # library default of the single-fit pass (csrc/xtv.hip dispatch_xtv)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 100 at N=1, 5 whole cv_iht runs at N>1)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=["auto", "fit", "cv", "replicas"], default="auto",
                    help="auto: fit (configs[2]) at N=1, cv (configs[3], strong-scaled) at N>1")
    ap.add_argument("--n", type=int, default=int(os.environ.get("MIH_BENCH_N", 500_000)))
    ap.add_argument("--p", type=int, default=int(os.environ.get("MIH_BENCH_P", 1_000_000)))
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--variant", type=int, default=-1, help="X'r kernel variant (-1 = library default)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of each baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cv", action="store_true", help="skip the secondary cv_iht measurement at N=1")
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
    pc = min(pc, max(probe, int(2e9 // ((n + 3) // 4))))    # keep the sample under ~2 GB of host memory
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
    return {"value": 1.0 / t_full, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"one whole oracle iht_one_step (iht_stepsize!, _iht_gradstep!, update_xb!, loglikelihood, score!) with k={kk} "
                      f"on the first {pc} of {p} SNP columns, n={n}: {t_step:.3f} s per step, of which the X'r pass {t_xtv:.3f} s "
                      f"(scaled by p/{pc}; the remaining {t_rest:.3f} s per step does not grow with p); {cores} OpenMP threads = the "
                      f"fastest of {sorted(trials)} on {ncpu} logical CPUs{'' if quota is None else f', cgroup CPU quota {quota:g}'}; "
                      "CPU restatement (oracle/), not MendelIHT.jl",
            "xtv_GBps": ((n + 3) // 4) * pc / t_xtv / 1e9, "step_s_on_sample": t_step, "xtv_s_on_sample": t_xtv}, cores


def cpu_baseline_cv(m, n, p, seed, cores, gpu_fits_per_s):
    """The oracle's cross-validation fits on a reduced grid of configs[3]: same rows, the first `pc` columns, 5 folds x 3
    model sizes, each fit on its training mask exactly as cv_iht runs it (cross_validation.jl:100-112).  Only the X'r
    passes grow with the column count: full-size time = passes * t_pass(sample) * p / pc + (measured time - passes *
    t_pass(sample)); fits/s at full size follows."""
    from oracle import oracle as O
    from conftest import hash_folds

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
    return {"value": v, "unit": "fits/s", "cores": cores, "kind": "port",
            "sample": f"oracle fits of cv_iht Bernoulli/Logit, 5 folds x path={path} ({fits} fits, {passes} X'r passes) on n={n} x the first "
                      f"{pc} of {p} columns: {dt:.2f} s, of which {passes * t_pass:.2f} s in X'r passes ({t_pass * 1e3:.1f} ms each; scaled by "
                      f"p/{pc}) and {rest:.2f} s in work that does not grow with p; {cores} OpenMP threads on {ncpu} logical CPUs"
                      f"{'' if quota is None else f', cgroup CPU quota {quota:g}'}; CPU restatement, not MendelIHT.jl",
            "gpu_over_cpu": (gpu_fits_per_s / v) if v > 0 else None,
            "note": "BASELINE's target is >= 20x over a 2-socket CPU on cv_iht path=1:20; this host exposes a 16-CPU quota of its 2 sockets"}


def cv_problem(m, x, n, p):
    from conftest import hash_folds
    rng = np.random.default_rng(2025)
    supp = np.sort(rng.choice(p, 10, replace=False))
    eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    return yb, hash_folds(n, 5)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        # `--gpus 8` without a launcher used to measure one GPU silently.  (Checked before any GPU call.)
        sys.exit(f"bench.py --gpus {a.gpus} needs {a.gpus} ranks but WORLD_SIZE={world}: launch it as\n"
                 f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 "
                 f"--master-port 29500 bench.py --gpus {a.gpus} [--steps K --warmup W]")
    mode = a.mode if a.mode != "auto" else ("fit" if world == 1 else "cv")
    if a.steps is None:
        a.steps = 100 if mode in ("fit", "replicas") else 5
    if a.warmup is None:
        a.warmup = 5 if mode in ("fit", "replicas") else 1
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # MIH_BENCH_BACKEND=gloo + MIH_BENCH_ONE_DEVICE=1: multi-rank smoke test on a single-GPU box
    backend = os.environ.get("MIH_BENCH_BACKEND", "nccl")
    if os.environ.get("MIH_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    import mendeliht_amd as m
    if not os.path.exists(m.library_path()):       # fresh checkout: compile the HIP library first
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if world > 1:
            dist.barrier()

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n, p, k = a.n, a.p, a.k
    if a.variant >= 0:
        m.lib().mih_set_xtv_variant(a.variant)
    seed = 2024 + (rank if mode == "replicas" else 0)       # cv: identical replicas of X on every rank
    t_gen = time.perf_counter()
    x = m.SnpLinAlg.synthetic(n, p, seed=seed, device=local)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    if mode == "cv":
        # ---- BASELINE configs[3], strong-scaled over the ranks ---------------------------------------------------------
        from mendeliht_amd import dist as D
        yb, folds = cv_problem(m, x, n, p)
        path = range(1, 21)

        def one_cv():
            t0 = time.perf_counter()
            _, raw = m.cv_iht(yb, x, None, path=path, q=5, folds=folds, verbose=False, return_raw=True, rank=rank, world=world,
                              d=m.Bernoulli(), l=m.LogitLink())
            t1 = time.perf_counter()
            tot = D.gather_losses(raw)                      # the path's one exchange: all-gather of the held-out losses
            return tot, t1 - t0, time.perf_counter() - t1
        for _ in range(a.warmup):
            one_cv()
        m.profile_read(reset=True)
        m.profile_enable(True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_fit = t_gather = 0.0
        for _ in range(a.steps):
            tot, tf, tg = one_cv()
            t_fit += tf
            t_gather += tg
        torch.cuda.synchronize()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        m.profile_enable(False)
        xtv_ms, launches = m.profile_read(reset=True)
        fits_mine = len(D.shard_combinations(5, 20, rank, world))
        stats = torch.tensor([launches / a.steps, xtv_ms / a.steps, 1e3 * t_gather / a.steps, fits_mine],
                             dtype=torch.float64, device="cuda" if backend == "nccl" and world > 1 else "cpu")
        if world > 1:
            allst = [torch.empty_like(stats) for _ in range(world)]
            dist.all_gather(allst, stats)
            allst = [s.cpu().tolist() for s in allst]
        else:
            allst = [stats.tolist()]
        if rank == 0:
            from conftest import hash_folds  # noqa: F401
            mse = np.zeros(20)
            ninfold = np.bincount(folds - 1, minlength=5)
            for j in range(5):
                mse += tot[j] * ninfold[j] / n               # meanloss (cross_validation.jl:304-320)
            passes = sum(s[0] for s in allst)
            # one fused pass scores up to 15 residuals; an IHT iteration of one fit = one score of one residual.  The
            # library counts launches, not residuals, so iterations are reported from the deterministic single-rank count
            out = {
                "metric": METRIC,
                "value": 100 * a.steps / elapsed, "unit": "fits/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                "config": {"workload": f"cv_iht Bernoulli/Logit path=1:20, 5 folds on synthetic SnpArray n={n} p={p} (BASELINE configs[3]): "
                                       f"100 (fold,k) fits per step, combination i on rank i mod {world}, identical X replica per GPU, "
                                       "ONE all-gather of the held-out losses per step inside the timed region "
                                       "(cross_validation.jl:98-121)",
                           "n": n, "p": p, "path": "1:20", "folds": 5, "generator_s": round(t_gen, 2), "best_k": int(np.argmin(mse)) + 1,
                           "cv_iht_s": elapsed / a.steps,
                           "compare_with": "the N=1 line's cv_iht.fits_per_s (= 100 / cv_iht.cv_iht_s): same workload on one GPU"},
                "per_rank": [{"rank": i, "fits": int(s[3]), "fused_passes_per_step": s[0], "xtv_kernel_ms_per_step": s[1],
                              "gather_ms_per_step": s[2]} for i, s in enumerate(allst)],
                "fused_passes_per_step_total": passes,
                "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None,
                             "note": "fused multi-RHS passes: see the N=1 line for the single-fit pass and DESIGN.md 3.1b"},
            }
            if launches:
                mres = 100.0 * 12.5 / max(passes, 1)         # ~12.5 residual scores per fit (1247 per 100 fits, deterministic at N=1)
                alg = x.algorithmic_bytes(12)
                kern_ms = sum(s[1] for s in allst) / max(passes, 1)
                out["roofline"].update(achieved=alg / (kern_ms * 1e-3) / 1e9, frac=alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                       kernel="k_xtv_dma16<NR,2,8,D> (mean over the fused passes of all ranks, 1 to 5 operands each; bytes as for 12 residuals)",
                                       kernel_ms=kern_ms, launches=int(passes * a.steps), algorithmic_bytes_per_launch=alg,
                                       residuals_per_pass_estimate=mres)
            print(json.dumps(out), flush=True)
        del x
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- BASELINE configs[2]: one fit, step = iht_one_step! ---------------------------------------------------------------
    # phenotype: y = X beta + 1 + N(0,1), k true effects ~ N(0,1)  (simulate_utilities.jl:215-228)
    rng = np.random.default_rng(2025 + (rank if mode == "replicas" else 0))
    supp = np.sort(rng.choice(p, size=k, replace=False))
    beta = rng.standard_normal(k)
    y = x.xv_sparse(supp, beta) + 1.0 + rng.standard_normal(n)

    sess = m.IHTSession(y, x, None, k=k, d=m.Normal(), l=m.IdentityLink())
    for _ in range(a.warmup):
        sess.step()
    m.profile_read(reset=True)
    m.profile_enable(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    logl, nbt, tol = sess.run(a.steps)          # EXACTLY a.steps iterations, looped inside the library (mih_session_run)
    torch.cuda.synchronize()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    m.profile_enable(False)
    xtv_ms, launches = m.profile_read(reset=True)

    bhat, _ = sess.model()
    recovered = int(np.intersect1d(np.flatnonzero(bhat), supp).size)
    sess.close()

    if rank == 0:
        traffic_src = None
        if a.traffic_bytes is None:      # PMC traffic comes from a separate rocprofv3 pass of this command (profiles/)
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))
                if t["workload"] == {"n": n, "p": p} and t["kernel"] == XTV_KERNEL and a.variant < 0:
                    a.traffic_bytes = t["hbm_bytes_per_launch"]
                    traffic_src = "profiles/r02_traffic.json (separate rocprofv3 --pmc passes of this command, kernel " + t["kernel"] + ")"
            except (OSError, KeyError, ValueError):
                pass
        else:
            traffic_src = "--traffic-bytes"
        alg_bytes = x.algorithmic_bytes(1)
        kern_ms = xtv_ms / max(launches, 1)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if launches else 0.0
        out = {
            "metric": METRIC,
            "value": world * a.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": f"iht on synthetic SnpArray n={n} p={p} k={k} Normal/Identity (BASELINE configs[2])"
                                   + (", one independent replica per GPU" if world > 1 else ""),
                       "n": n, "p": p, "k": k, "xtv_variant": a.variant, "generator_s": round(t_gen, 2),
                       "backtracks_in_timed_steps": nbt, "true_effects_recovered": f"{recovered}/{k}",
                       "final_logl": logl, "host_and_small_kernels_ms_per_step": 1e3 * elapsed / a.steps - kern_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": a.traffic_bytes, "traffic_source": traffic_src,
                         "kernel": XTV_KERNEL if a.variant < 0 else "k_xtv_mfma", "kernel_ms": kern_ms, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        gpu_fits_per_s = None
        if world == 1 and not a.no_cv:
            # configs[3] on this one GPU (the N>1 mode's workload at N=1), same matrix
            yb, folds = cv_problem(m, x, n, p)
            m.profile_read(reset=True)
            m.profile_enable(True)
            t0 = time.perf_counter()
            mse, raw = m.cv_iht(yb, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True,
                                d=m.Bernoulli(), l=m.LogitLink())
            dt = time.perf_counter() - t0
            m.profile_enable(False)
            cms, cl = m.profile_read(reset=True)
            gpu_fits_per_s = 100.0 / dt
            out["cv_iht"] = {"workload": "cv_iht Bernoulli/Logit path=1:20, 5 folds (BASELINE configs[3]), all 100 fits on this GPU",
                             "seconds": dt, "cv_iht_s": dt, "fits": int(np.count_nonzero(raw)), "fits_per_s": gpu_fits_per_s, "best_k": int(np.argmin(mse)) + 1,
                             "fused_passes": int(cl), "xtv_kernel_ms_total": cms}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], cores = cpu_baseline(m, n, p, k, seed, a.cpu_seconds)
            if gpu_fits_per_s is not None:
                out["cpu_baseline_cv"] = cpu_baseline_cv(m, n, p, seed, cores, gpu_fits_per_s)
        print(json.dumps(out), flush=True)
    del x
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
