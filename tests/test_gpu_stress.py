"""Race hunting by repetition.  GPU sanitizers are not available on the pool, so the only detector of an ordering
bug in a cross-workgroup protocol (DESIGN.md 3.4's table: gate word, tickets, pinned record, spin flag) is to run the same work many
times and demand ONE set of bits.  Round 5 found such a bug by luck (a workgroup-scope fence in front of k_res_stats's ticket that
compiled to nothing: a stale partial in one run of three).  Every workload below runs hundreds of times on ONE matrix handle, then
from four host threads at once; tools/stress_chain.py is the same loop for 10^4 repetitions under gpurun (profiles/)."""
import hashlib
import os
import threading
import time

import numpy as np
import pytest

from conftest import hash_folds

pytestmark = pytest.mark.gpu


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())
    return h.hexdigest()


def workloads(mih, normal_data):
    """name -> (callable returning a digest of everything the run produced, repetitions in the suite)"""
    n = normal_data["n"]
    x = mih.SnpLinAlg(mih.read_bed(normal_data["bed"], n), n, center=True, scale=True, impute=True)
    y, z = normal_data["y"], normal_data["z"]
    xm = mih.SnpLinAlg.synthetic(6001, 900, seed=3, missing_rate=0.01)
    rng = np.random.default_rng(20260)
    supp = np.sort(rng.choice(900, 8, replace=False))
    em = xm.xv_sparse(supp, rng.standard_normal(8) * 0.6)
    zm = np.column_stack([np.ones(6001), rng.standard_normal(6001)])
    yp = rng.poisson(np.exp(0.25 * em + 0.1 * zm[:, 1])).astype(float)
    yp[17] = 400.0                                   # (the outlier side channel of the fixed-point residual fires in the first steps)
    yn = em + 0.5 + rng.standard_normal(6001)
    Y3 = np.vstack([yn, 0.5 * yn + rng.standard_normal(6001), rng.standard_normal(6001)])
    folds = hash_folds(6001, 3)

    def g1():                                        # the reference's recorded run, resident steps (gate word, tickets, pinned record)
        r = mih.fit_iht(y, x, z, k=7, verbose=False)
        return digest(r.beta, r.c, r.trace["logl"], r.trace["tol"], r.trace["backtracks"], r.mu, [r.iter])

    def poisson():                                   # 6 001 rows, imputed entries, backtracking: attempt slots, series that go on, k_res_peel
        r = mih.fit_iht(yp, xm, zm, k=8, d=mih.Poisson(), l=mih.LogLink(), verbose=False, max_iter=40)
        assert int(np.sum(r.trace["backtracks"])) > 0
        return digest(r.beta, r.c, r.trace["logl"], r.trace["tol"], r.trace["backtracks"], r.mu, [r.iter])

    def cv():                                        # 3 folds x 8 model sizes: two lanes' coroutines, spin-flag readbacks, fused passes
        _, raw = mih.cv_iht(yn, xm, zm, path=range(1, 9), q=3, folds=folds, verbose=False, return_raw=True)
        return digest(raw)

    def mv():                                        # three traits
        r = mih.fit_iht(Y3, xm, None, k=12, verbose=False, max_iter=15)
        return digest(r.beta, r.c, r.trace["logl"], [r.iter])

    return {"g1": (g1, 2000), "poisson": (poisson, 2000), "cv": (cv, 300), "mv": (mv, 1000)}, (x, xm)


def test_repeated_runs_give_one_set_of_bits(mih, normal_data):
    """Each workload N times in a row on one handle (2000 x the G1 fit, 2000 x the Poisson fit, 300 x the 24-fit cross-validation,
    1000 x the three-trait fit): one digest.  About a minute; a workload that is slower than expected on the box is cut at its
    share of the time (at least a quarter of its repetitions must have run)."""
    work, keep = workloads(mih, normal_data)
    report = {}
    for name, (fn, reps) in work.items():
        first = fn()
        t0, done = time.perf_counter(), 1
        while done < reps and time.perf_counter() - t0 < 20.0:
            assert fn() == first, (name, done)
            done += 1
        report[name] = (done, round(time.perf_counter() - t0, 1))
        assert done >= reps // 4, (name, report)
    print("stress repetitions (count, seconds):", report)
    del keep


def test_four_host_threads_give_the_same_bits(mih, normal_data):
    """The same workloads from four host threads at once on the SAME two handles (cross_validation.jl:100-112 calls the path
    concurrently on a shared x): every result equals the single-threaded digest."""
    work, keep = workloads(mih, normal_data)
    want = {name: fn() for name, (fn, _) in work.items()}
    errs, counts = [], {name: 0 for name in work}
    lock = threading.Lock()
    stop_at = time.perf_counter() + 20.0
    order = [["g1", "poisson", "mv", "cv"], ["poisson", "g1", "cv", "mv"], ["mv", "cv", "g1", "poisson"], ["cv", "mv", "poisson", "g1"]]

    def run(tid):
        try:
            rounds = 0
            while time.perf_counter() < stop_at and rounds < 200:
                for name in order[tid]:
                    got = work[name][0]()
                    if got != want[name]:
                        raise AssertionError((tid, name, rounds))
                    with lock:
                        counts[name] += 1
                rounds += 1
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert min(counts.values()) >= 8, counts
    print("concurrent stress repetitions:", counts)
    del keep
