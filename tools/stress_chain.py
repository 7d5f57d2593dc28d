#!/usr/bin/env python3
"""tools/stress_chain.py [repetitions] -- the repetition loop of tests/test_gpu_stress.py at 10^4 repetitions per workload mix
(VERDICT r5 item 3): G1 resident fit, a 6 001-row Poisson fit with backtracking and a planted count outlier, a 3 x 8 cross-validation
and a three-trait fit, on ONE pair of handles, every run's digest compared with the first.  Prints one JSON line.
    gpurun --timeout 3000 -- 'python tools/stress_chain.py 10000 > gpurun_out/stress.json'"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np                     # noqa: E402
import mendeliht_amd as mih            # noqa: E402
import test_gpu_stress as T            # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 2400.0
    FIX = os.path.join(ROOT, "tests", "fixtures")
    n = 1000
    y = np.loadtxt(os.path.join(FIX, "normal_y_fam6.txt"))
    z = np.loadtxt(os.path.join(FIX, "covariates.txt"), delimiter=",")
    mu = z[:, 1:].mean(axis=0)
    z[:, 1:] = (z[:, 1:] - mu) / np.sqrt(((z[:, 1:] - mu) ** 2).sum(axis=0) / (n - 1))
    data = dict(n=n, bed=os.path.join(FIX, "normal.bed"), y=y, z=z)
    work, keep = T.workloads(mih, data)
    # shares of the repetitions: the cheap chains most often (they turn the gate / ticket / record protocols over fastest)
    share = {"g1": 0.45, "poisson": 0.35, "mv": 0.15, "cv": 0.05}
    first = {name: fn() for name, (fn, _) in work.items()}
    counts = {name: 0 for name in work}
    mismatches = []
    t0 = time.perf_counter()
    for name, (fn, _) in work.items():
        reps = int(total * share[name])
        t1 = time.perf_counter()
        for i in range(reps):
            if fn() != first[name]:
                mismatches.append((name, i))
            counts[name] += 1
            if time.perf_counter() - t0 > budget:
                break
        counts[name + "_s"] = round(time.perf_counter() - t1, 1)
    print(json.dumps({"repetitions": counts, "mismatches": mismatches, "seconds": round(time.perf_counter() - t0, 1),
                      "digests": first}))
    return 1 if mismatches else 0


if __name__ == "__main__":
    sys.exit(main())
