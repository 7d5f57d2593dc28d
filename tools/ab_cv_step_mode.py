#!/usr/bin/env python3
"""tools/ab_cv_step_mode.py [rounds] -- cv_iht at BASELINE configs[3] size (100 Bernoulli/Logit fits, n = 500k, p = 1M) with the lanes'
fits stepping resident on the device (mih_fit_params::step_mode 0, round 6) against host-driven steps (1, rounds 1-5): ONE process,
one matrix, the two modes alternating; the 5 x 20 losses bit for bit, the lock-step counters of either mode.  Prints JSON lines."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mendeliht_amd as m

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, p = 500_000, int(os.environ.get("MIH_P", 1_000_000))
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = m.hash_folds(n, 5)
digits = int(os.environ.get("MIH_DIGITS", "0"))


def run(mode, hook):
    m.set_step_mode(mode)
    if hook:
        m.profile_read(x, reset=True); m.profile_counters(x, reset=True); m.profile_enable(x, True)
    t0 = time.perf_counter()
    mse, raw = m.cv_iht(y, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, return_raw=True, d=m.Bernoulli(), l=m.LogitLink(),
                        xtv_digits=digits)
    dt = time.perf_counter() - t0
    out = {"step_mode": mode, "seconds": round(dt, 4), "best_k": int(np.argmin(mse)) + 1, "hash": hashlib.sha256(raw.tobytes()).hexdigest()[:16]}
    if hook:
        m.profile_enable(x, False)
        ps = m.profile_passes(x, reset=True)
        c = m.profile_counters(x, reset=True)
        out.update(passes=len(ps), busy_union_ms=round(m.busy_union_ms(ps), 1), pass_ms_sum=round(sum(q["ms"] for q in ps), 1),
                   resident_steps=c["resident_steps"], resident_attempts=c["resident_attempts"], resident_redos=c["resident_redos"],
                   handbacks=c["resident_handbacks"], scores=c["scores"], rounds=c["rounds"])
    return out


run(0, False); run(1, False)                  # warm-up of both paths
best = {0: 1e9, 1: 1e9}
hashes = set()
for rnd in range(rounds):
    for mode in (1, 0):
        o = run(mode, rnd == 0)
        best[mode] = min(best[mode], o["seconds"])
        hashes.add(o["hash"])
        print(json.dumps(o), flush=True)
print(json.dumps({"min_seconds_host_driven": best[1], "min_seconds_resident": best[0], "losses_bit_identical": len(hashes) == 1}))
m.set_step_mode(0)
sys.exit(0 if len(hashes) == 1 else 1)
