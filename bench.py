#!/usr/bin/env python3
"""bench.py -- IHT iterations/s and X'r GB/s vs the HBM roofline (BASELINE.json metric).

A "step" is one IHT iteration (iht_one_step!, src/fit.jl:213-263) over a synthetic 2-bit
SnpArray resident in HBM: step size (k-column X v), gradient step + top-k projection,
X beta, mean / loglikelihood, backtracking if needed, and the full X'r score pass.
N = 1 runs BASELINE configs[2] (n=500k, p=1M, k=200, Normal).  N > 1 runs one independent
replica per rank (the path shards only across independent fits -- weak scaling, no
data-path collective; the driver launches ranks with torch.distributed.run).

Prints ONE JSON line (rank 0).  `roofline` is measured live: HIP events around every
launch of the dominant kernel (k_xtv_mfma_lds, the X'r pass) inside the timed region, on the stream it runs on.
`cpu_baseline` times the CPU oracle (a port, not MendelIHT.jl itself -- no Julia in the
image) on a bounded column sample of the same matrix, rank 0 at N = 1 only.
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

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=int(os.environ.get("MIH_BENCH_N", 500_000)))
    ap.add_argument("--p", type=int, default=int(os.environ.get("MIH_BENCH_P", 1_000_000)))
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--variant", type=int, default=-1, help="X'r kernel variant (-1 = library default)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per k_xtv launch from a separate rocprofv3 --pmc pass (corrected)")
    return ap.parse_args()


def cpu_baseline(m, n, p, seed, target_s):
    """Time the oracle's X'r (OpenMP over column blocks) on the first columns of the same synthetic matrix;
    returns the JSON object.  The thread count is tuned on a probe sample first: on a box whose container has
    a CPU quota, or two sockets, more threads than that are slower, and `cores` reports what was used."""
    from oracle import oracle as O

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None                                            # cgroup v2 CPU quota of the container, in CPUs
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
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
    pc = int(min(p, max(probe, probe * target_s / max(t, 1e-6))))
    pc = min(pc, max(probe, int(2e9 // ((n + 3) // 4))))    # keep the sample under ~2 GB of host memory
    ox = load(pc)
    t = time_pass(ox, target_s / 3.0)
    per_col = t / pc
    iters_per_s = 1.0 / (per_col * p)
    return {"value": iters_per_s, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"oracle X'r (one IHT iteration = one pass) on the first {pc} of {p} SNP columns, "
                      f"n={n}, {t:.3f} s per pass (mean over >= {target_s / 3.0:.0f} s of repeats, {cores} OpenMP threads = the "
                      f"fastest of {sorted(trials)} on {ncpu} logical CPUs{'' if quota is None else f', cgroup CPU quota {quota:g}'}), scaled by p/{pc}; "
                      "CPU restatement, not MendelIHT.jl",
            "xtv_GBps": ((n + 3) // 4) * pc / t / 1e9}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    n, p, k = a.n, a.p, a.k
    if a.variant >= 0:
        m.lib().mih_set_xtv_variant(a.variant)
    seed = 2024 + rank
    t_gen = time.perf_counter()
    x = m.SnpLinAlg.synthetic(n, p, seed=seed, device=local)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    # phenotype: y = X beta + 1 + N(0,1), k true effects ~ N(0,1)  (simulate_utilities.jl:215-228)
    rng = np.random.default_rng(2025 + rank)
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
    logl = bt = tol = None
    nbt = 0
    for _ in range(a.steps):
        logl, bt, tol = sess.step()
        nbt += bt
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    m.profile_enable(False)
    xtv_ms, launches = m.profile_read(reset=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    bhat, _ = sess.model()
    recovered = int(np.intersect1d(np.flatnonzero(bhat), supp).size)
    sess.close()

    if rank == 0:
        if a.traffic_bytes is None:      # PMC traffic comes from a separate rocprofv3 pass (profiles/)
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
                if t["workload"] == {"n": n, "p": p}:
                    a.traffic_bytes = t["hbm_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                pass
        alg_bytes = x.algorithmic_bytes(1)
        kern_ms = xtv_ms / max(launches, 1)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if launches else 0.0
        out = {
            "metric": "IHT iterations/sec + X'r GB/s vs HBM roofline, n=500k p=1M k=200",
            "value": world * a.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"iht on synthetic SnpArray n={n} p={p} k={k} Normal/Identity (BASELINE configs[2]), "
                                   "one independent replica per GPU",
                       "n": n, "p": p, "k": k, "xtv_variant": a.variant, "generator_s": round(t_gen, 2),
                       "backtracks_in_timed_steps": nbt, "true_effects_recovered": f"{recovered}/{k}",
                       "final_logl": logl},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": a.traffic_bytes,
                         "kernel": "k_xtv_mfma_lds<1,1,4>" if a.variant < 0 else "k_xtv_mfma", "kernel_ms": kern_ms, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, n, p, seed, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    del x
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
