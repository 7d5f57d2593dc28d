#!/usr/bin/env python3
"""bench.py -- IHT iterations/s and X'r GB/s vs the HBM roofline (BASELINE.json metric).

N = 1 (default): BASELINE configs[2].  A "step" is one IHT iteration (iht_one_step!, src/fit.jl:213-263) over a
synthetic 2-bit SnpArray resident in HBM (n=500k, p=1M, k=200, Normal): step size (k-column X v), gradient step +
top-k projection, X beta, mean / loglikelihood, backtracking if needed, and the full X'r score pass.  `roofline` is
measured live: HIP events around every launch of the dominant kernel (the X'r pass) inside the timed region, on the
stream it runs on, recorded by the library together with the NAME of the kernel it dispatched and the number of residuals
the launch scored (mih_profile_passes).  Secondary objects in the same line: `cv_iht` (BASELINE configs[3] --
Bernoulli/Logit, path=1:20, 5 folds, all 100 fits -- on this one GPU, same matrix), `cpu_baseline` (the CPU oracle's
whole iht_one_step on a bounded column sample of the same matrix) and `cpu_baseline_cv` (the oracle's cv_iht on a reduced
grid of configs[3]).

N > 1: the path's real shard, BASELINE configs[3] -- cross_validation.jl:98-121.  `python bench.py --gpus N` starts its N
ranks itself (a child `python -m torch.distributed.run`, before anything touches a GPU) and relays rank 0's line; started
under a launcher (WORLD_SIZE set) it is one of the ranks.  Every rank holds an identical replica of X; a step is ONE whole
cv_iht (100 (fold,k) fits) strong-scaled over the ranks with the single RCCL all-gather of the held-out losses inside the
timed region.  `value` keeps the metric's unit at every N: IHT iterations (= residual scores, counted by the library,
summed over the fits of all ranks) per second; fits/s, passes per rank and the gather time are reported beside it.
`--mode replicas` keeps round 1's independent replicas of configs[2].

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
METRIC = "IHT iterations/sec + X'r GB/s vs HBM roofline, n=500k p=1M k=200"
DTYPE = "f64 (residual as a 54-bit fixed-point number, exact accumulation on the matrix cores, f64 recombination)"
TRAFFIC_FILE = os.path.join("profiles", "r03_traffic.json")    # separate rocprofv3 --pmc passes of this command (tools/prof_bench.sh)


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
    ap.add_argument("--variant", type=int, default=-1, help="measurement build only (MENDELIHT_HIP_PROBES=1): a round-1 per-wave kernel shape")
    ap.add_argument("--dry-run", action="store_true", help="print each rank's launcher environment and exit (no GPU call)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-started ranks (0 = a free one)")
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


def cores_used(threads):
    """`cores` of a CPU baseline = the CPUs that actually ran it: min(OpenMP threads, cgroup quota)."""
    _, quota = host_cpus()
    return threads if quota is None else int(min(threads, max(1, round(quota))))


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `python -m torch.distributed.run` BEFORE this
    process touches a GPU (it never does), relay the child's stdout (rank 0's JSON line) and return its exit code."""
    if not a.dry_run and not os.environ.get("MIH_BENCH_ONE_DEVICE"):
        import torch                                             # counting devices does not initialise the GPU
        have = torch.cuda.device_count()
        if have < a.gpus:
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
    return {"value": 1.0 / t_full, "unit": "iterations/s", "cores": cores_used(cores), "omp_threads": cores, "kind": "port",
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


def pass_stats(m, x, passes):
    """Per-launch records of the dominant kernel -> what the roofline object needs: launches, summed and union busy time,
    ALGORITHMIC bytes of exactly those launches (algorithmic_bytes(residuals) each), residual scores, the kernel's name."""
    if not passes:
        return {"launches": 0, "ms_sum": 0.0, "ms_union": 0.0, "bytes": 0.0, "residuals": 0, "kernel": None, "kernels": {}}
    by_m = {}
    names = {}
    for q in passes:
        by_m[q["residuals"]] = by_m.get(q["residuals"], 0) + 1
        names[q["kernel"]] = names.get(q["kernel"], 0) + 1
    alg = {mm: x.algorithmic_bytes(mm) for mm in by_m}
    return {"launches": len(passes), "ms_sum": sum(q["ms"] for q in passes), "ms_union": m.busy_union_ms(passes),
            "bytes": float(sum(alg[q["residuals"]] for q in passes)), "residuals": int(sum(q["residuals"] for q in passes)),
            "kernel": max(names, key=names.get), "kernels": names}


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
                                     "n_gpus": a.gpus}) + "\n")      # one write per rank: the ranks share the launcher's pipe
        sys.stdout.flush()
        return
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
        m.probe_set(variant=a.variant)                      # raises unless the measurement build is loaded
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
        m.profile_read(x, reset=True)
        m.profile_counters(x, reset=True)
        m.profile_enable(x, True)
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
        m.profile_enable(x, False)
        st = pass_stats(m, x, m.profile_passes(x, reset=True))
        cnt = m.profile_counters(x, reset=True)
        stats = torch.tensor([st["launches"], st["ms_sum"], st["ms_union"], st["bytes"], st["residuals"], cnt["scores"], cnt["fits"],
                              1e3 * t_gather, 1e3 * t_fit],
                             dtype=torch.float64, device="cuda" if backend == "nccl" and world > 1 else "cpu")
        if world > 1:
            allst = [torch.empty_like(stats) for _ in range(world)]
            dist.all_gather(allst, stats)
            allst = [q.cpu().tolist() for q in allst]
            names = [None] * world
            dist.all_gather_object(names, st["kernels"])
        else:
            allst = [stats.tolist()]
            names = [st["kernels"]]
        if rank == 0:
            mse = np.zeros(20)
            ninfold = np.bincount(folds - 1, minlength=5)
            for j in range(5):
                mse += tot[j] * ninfold[j] / n               # meanloss (cross_validation.jl:304-320)
            K = a.steps
            launches = sum(q[0] for q in allst)
            ms_sum = sum(q[1] for q in allst)
            nbytes = sum(q[3] for q in allst)
            scores = sum(q[5] for q in allst)               # IHT iterations incl. the initial score of every fit, all ranks, all steps
            fits = sum(q[6] for q in allst)
            kern = {}
            for d_ in names:
                for kn, c in d_.items():
                    kern[kn] = kern.get(kn, 0) + c
            out = {
                "metric": METRIC,
                "value": scores / elapsed, "unit": "iterations/s",
                "n_gpus": world, "steps": K, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / K,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                "config": {"workload": f"cv_iht Bernoulli/Logit path=1:20, 5 folds on synthetic SnpArray n={n} p={p} (BASELINE configs[3]): "
                                       f"100 (fold,k) fits per step sharded over {world} ranks (mih_cv_assignment), identical X replica per GPU, "
                                       "ONE all-gather of the held-out losses per step inside the timed region "
                                       "(cross_validation.jl:98-121); value = IHT iterations of all fits per second "
                                       "(an iteration = one residual score; up to 15 fits share one fused X'R pass)",
                           "n": n, "p": p, "path": "1:20", "folds": 5, "generator_s": round(t_gen, 2), "best_k": int(np.argmin(mse)) + 1,
                           "cv_iht_s": elapsed / K, "fits_per_s": fits / elapsed, "fits_per_step": fits / K,
                           "iterations_per_step": scores / K,
                           "compare_with": "the N=1 line's cv_iht object (same workload on one GPU: cv_iht.iterations_per_s, "
                                           "cv_iht.fits_per_s); the N=1 `value` is the single-fit workload configs[2]"},
                "per_rank": [{"rank": i, "fits_per_step": q[6] / K, "iterations_per_step": q[5] / K, "fused_passes_per_step": q[0] / K,
                              "xtv_kernel_ms_per_step": q[1] / K, "xtv_busy_union_ms_per_step": q[2] / K,
                              "cv_iht_ms_per_step": q[8] / K, "gather_ms_per_step": q[7] / K} for i, q in enumerate(allst)],
                "fused_passes_per_step_total": launches / K,
                "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None},
            }
            if launches:
                out["roofline"].update(
                    achieved=nbytes / (ms_sum * 1e-3) / 1e9, frac=nbytes / (ms_sum * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    kernel=max(kern, key=kern.get), kernels=kern, kernel_ms=ms_sum / launches, launches=int(launches),
                    algorithmic_bytes_per_launch=nbytes / launches, residuals_per_launch=sum(q[4] for q in allst) / launches,
                    note="measured: sum over the launches of all ranks of algorithmic_bytes(residuals of that launch) / sum of their "
                         "HIP-event durations; the fused passes carry 1 to 15 residuals each (kernels = launches per dispatched kernel)")
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
    m.profile_read(x, reset=True)
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
    xtv_ms, launches = st["ms_sum"], st["launches"]

    bhat, _ = sess.model()
    recovered = int(np.intersect1d(np.flatnonzero(bhat), supp).size)
    sess.close()

    if rank == 0:
        traffic_src = None
        if a.traffic_bytes is None:      # PMC traffic comes from a separate rocprofv3 pass of this command (profiles/)
            try:
                t = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
                if t["workload"] == {"n": n, "p": p} and t["kernel"] == st["kernel"]:
                    a.traffic_bytes = t["hbm_bytes_per_launch"]
                    traffic_src = (f"{TRAFFIC_FILE} (separate rocprofv3 --pmc passes of this command, kernel {t['kernel']}; "
                                   "not collected by this run)")
            except (OSError, KeyError, ValueError):
                pass
        else:
            traffic_src = "--traffic-bytes"
        alg_bytes = st["bytes"] / launches if launches else x.algorithmic_bytes(1)
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
                         "kernel": st["kernel"], "kernel_ms": kern_ms, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        gpu_fits_per_s = None
        if world == 1 and not a.no_cv:
            # configs[3] on this one GPU (the N>1 mode's workload at N=1), same matrix
            yb, folds = cv_problem(m, x, n, p)
            m.profile_read(x, reset=True)
            m.profile_counters(x, reset=True)
            m.profile_enable(x, True)
            t0 = time.perf_counter()
            mse, raw = m.cv_iht(yb, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True,
                                d=m.Bernoulli(), l=m.LogitLink())
            dt = time.perf_counter() - t0
            m.profile_enable(x, False)
            cst = pass_stats(m, x, m.profile_passes(x, reset=True))
            cnt = m.profile_counters(x, reset=True)
            gpu_fits_per_s = 100.0 / dt
            out["cv_iht"] = {"workload": "cv_iht Bernoulli/Logit path=1:20, 5 folds (BASELINE configs[3]), all 100 fits on this GPU",
                             "seconds": dt, "cv_iht_s": dt, "fits": int(np.count_nonzero(raw)), "fits_per_s": gpu_fits_per_s,
                             "iterations": cnt["scores"], "iterations_per_s": cnt["scores"] / dt, "best_k": int(np.argmin(mse)) + 1,
                             "fused_passes": cst["launches"], "residuals_scored_by_passes": cst["residuals"],
                             "xtv_busy_union_ms": cst["ms_union"], "xtv_kernel_ms_sum_over_lanes": cst["ms_sum"],
                             "xtv_kernel_ms_per_residual_scored": cst["ms_sum"] / max(cst["residuals"], 1),
                             "lockstep": cnt, "kernels": cst["kernels"],
                             "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                                          "achieved": cst["bytes"] / (cst["ms_sum"] * 1e-3) / 1e9 if cst["launches"] else None,
                                          "frac": cst["bytes"] / (cst["ms_sum"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if cst["launches"] else None,
                                          "kernel": cst["kernel"], "kernel_ms": cst["ms_sum"] / max(cst["launches"], 1),
                                          "algorithmic_bytes_per_launch": cst["bytes"] / max(cst["launches"], 1),
                                          "note": "sum of algorithmic_bytes(residuals of the launch) / sum of HIP-event durations; the two "
                                                  "lock-step lanes' passes overlap, so the sum exceeds the union (and may exceed the wall time). "
                                                  "A fused pass streams X once for up to 18 residuals and is bound by the matrix pipe under the "
                                                  "power cap, not by HBM: wider passes LOWER this fraction while the cost per residual falls "
                                                  "(xtv_kernel_ms_per_residual_scored; DESIGN.md 3.1b)"}}
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
