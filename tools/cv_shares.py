"""All eight `rank = r, world = 8` shares of BASELINE configs[3] (cv_iht Bernoulli/Logit, path = 1:20, 5 folds, n = 500k, p = 1M)
measured ONE AFTER THE OTHER on a single GPU: seconds, fused passes, fits and IHT iterations per rank, max / mean -- a
single-GPU PROJECTION of the 8-GPU run (what each GPU would do if the GPUs do not disturb each other), for both sharding rules
(MENDELIHT_CV_ASSIGN=0: round 2's fold-major `index mod world`; 1: round-robin over the combinations sorted by model size).
usage: cv_shares.py > profiles/r03_cv_shares.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # the sharding-rule switch is an A/B knob of the measurement build
import mendeliht_amd as m
n, p = int(os.environ.get("MIH_N", 500_000)), int(os.environ.get("MIH_P", 1_000_000))
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = m.hash_folds(n, 5)
kw = dict(path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink())
m.cv_iht(yb, x, None, rank=0, world=8, **kw)            # warm-up
out = {"_what": __doc__.split("usage")[0].strip(), "n": n, "p": p, "rules": {}}
ref = None
for rule in ("0", "1"):
    os.environ["MENDELIHT_CV_ASSIGN"] = rule
    best = None
    for rep in range(2):
        ranks, tot = [], None
        for r in range(8):
            m.profile_read(x, reset=True); m.profile_counters(x, reset=True); m.profile_enable(x, True)
            t0 = time.perf_counter()
            _, raw = m.cv_iht(yb, x, None, rank=r, world=8, **kw)
            dt = time.perf_counter() - t0
            m.profile_enable(x, False)
            ms, launches = m.profile_read(x, reset=True)
            c = m.profile_counters(x, reset=True)
            ranks.append({"rank": r, "seconds": dt, "fused_passes": launches, "xtv_kernel_ms": ms, "fits": c["fits"], "iterations": c["scores"],
                          "rounds": c["rounds"]})
            tot = raw if tot is None else tot + raw
        if ref is None:
            ref = tot.copy()
        assert np.array_equal(tot.view(np.uint64), ref.view(np.uint64)), "the losses depend on the sharding rule"
        secs = [q["seconds"] for q in ranks]
        run = {"ranks": ranks, "max_s": max(secs), "mean_s": float(np.mean(secs)), "max_over_mean": max(secs) / float(np.mean(secs)),
               "sum_s": float(sum(secs))}
        print(f"rule {rule} rep {rep}: max {run['max_s']:.3f} s mean {run['mean_s']:.3f} s max/mean {run['max_over_mean']:.3f}", file=sys.stderr, flush=True)
        if best is None or run["max_s"] < best["max_s"]:
            best = run
    out["rules"]["index_mod_world (round 2)" if rule == "0" else "k-stratified round-robin (library default)"] = best
m.profile_read(x, reset=True); m.profile_counters(x, reset=True)
os.environ["MENDELIHT_CV_ASSIGN"] = "1"
t0 = time.perf_counter()
_, raw = m.cv_iht(yb, x, None, rank=0, world=1, **kw)
out["single_gpu_all_100_fits_s"] = time.perf_counter() - t0
assert np.array_equal(raw.view(np.uint64), ref.view(np.uint64))
out["losses_bit_identical_across_rules_and_to_single_rank"] = True
d = out["rules"]["k-stratified round-robin (library default)"]
out["projected_8gpu_speedup_over_1gpu"] = out["single_gpu_all_100_fits_s"] / d["max_s"]
print(json.dumps(out, indent=1))
