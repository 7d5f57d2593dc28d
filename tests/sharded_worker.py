"""Worker of the column-sharded fit test: launched once per rank by torch.distributed.run
(tests/test_gpu_sharded.py).  Every rank builds its block of SNP columns on the SAME GPU (the test box
has one), the ranks talk over gloo; rank 0 also runs the unsharded fit and writes both to JSON."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mendeliht_amd as m                                   # noqa: E402
from mendeliht_amd import dist as D                         # noqa: E402
from conftest import FIX, make_bed                          # noqa: E402


def summary(res):
    nz = np.flatnonzero(res.beta)
    return dict(support=nz.tolist(), beta=res.beta[nz].tolist(), c=np.asarray(res.c).tolist(), logl=res.logl,
                iter=int(res.iter), logl_trace=res.trace["logl"].tolist(), bt=res.trace["backtracks"].tolist(),
                tol=res.trace["tol"].tolist(), choose_fired=bool(res.choose_fired), sigma_g=res.σg)


def summary_mv(res):
    """a multivariate fit (mIHTResult): beta is B, r x p"""
    B = np.asarray(res.beta)
    tr, col = np.nonzero(B)
    order = np.lexsort((tr, col))
    return dict(support=[[int(tr[t]), int(col[t])] for t in order], beta=[float(B[tr[t], col[t]]) for t in order],
                c=np.asarray(res.c).ravel(order="F").tolist(), logl=res.logl, iter=int(res.iter), logl_trace=res.trace["logl"].tolist(),
                bt=res.trace["backtracks"].tolist(), tol=res.trace["tol"].tolist(), choose_fired=bool(res.choose_fired),
                Sigma=np.asarray(res.Sigma).ravel().tolist(), sigma_g=np.asarray(res.sigma_g).tolist())


def mv_traits(x, rng, r, k, q):
    """r correlated traits with k planted effects and q covariates on the matrix x (GPU product: the worker has no CPU checker)"""
    p, n = x.p, x.n
    XB = np.zeros((r, n))
    for _ in range(k):
        XB[rng.integers(r)] += x.xv_sparse(np.array([rng.integers(p)]), np.array([rng.standard_normal() * 0.6]))
    A = rng.standard_normal((r, r))
    L = np.linalg.cholesky(A @ A.T / r + np.eye(r) * 0.5)
    Z = np.vstack([np.ones(n)] + [rng.standard_normal(n) for _ in range(q - 1)])
    return XB + rng.standard_normal((r, q)) @ Z + L @ rng.standard_normal((r, n)), Z


def native_main(out_path):
    """MIH_NATIVE=1: the library's own exchange (mih_comm_create_rccl: ncclAllReduce / ncclAllGather on a private stream) against
    the torch.distributed callbacks, rank by rank, bit for bit.  With one GPU per rank that is the real librccl over
    xGMI; MIH_NATIVE_ONE_DEVICE=1 (the one-GPU test box): every rank on device 0, torch.distributed over gloo, and the library
    loads the test-only stand-in tests/libfake_rccl.so through MENDELIHT_RCCL_LIB (set by the test) -- the same comm.hip code
    with more than one rank.  The callbacks sum in rank order (ordered_sum) as the stand-in does, so three ranks agree bit for bit."""
    one_dev = bool(os.environ.get("MIH_NATIVE_ONE_DEVICE"))
    rank, world, local = D.init_from_env(backend="gloo" if one_dev else "nccl")
    if one_dev:
        local = 0
    import torch.distributed as dist
    n = 1000
    cols = m.read_bed(os.path.join(FIX, "normal.bed"), n)
    y = np.loadtxt(os.path.join(FIX, "normal_y_fam6.txt"))
    z = np.loadtxt(os.path.join(FIX, "covariates.txt"), delimiter=",")
    z[:, 1:] = (z[:, 1:] - z[:, 1:].mean(axis=0)) / z[:, 1:].std(axis=0, ddof=1)
    rng = np.random.default_rng(42)
    n2, p2 = 1501, 2300
    cols2 = make_bed(rng, n2, p2, missing_rate=0.02)
    x2 = m.SnpLinAlg(cols2, n=n2, center=True, scale=True, impute=True, device=local)
    supp = np.sort(rng.choice(p2, 8, replace=False))
    eta = x2.xv_sparse(supp, rng.standard_normal(8) * 0.7)
    yb = (rng.random(n2) < 1 / (1 + np.exp(-eta))).astype(float)
    out = {}
    for name, (cc, nn, yy, zz, kw) in {"normal_k7": (cols, n, y, z, dict(k=7)),
                                       "logistic": (cols2, n2, yb, None, dict(k=9, d=m.Bernoulli(), l=m.LogitLink())),
                                       "init_beta": (cols2, n2, eta + 0.3 + np.random.default_rng(5).standard_normal(n2), None, dict(k=8, init_beta=True)),
                                       # (round 6) a count outlier: k_res_peel in the resident sharded chain, k_r_stats's guard in the callbacks' steps
                                       "poisson_outlier": (cols2, n2, np.where(np.arange(n2) == 77, 400.0, np.random.default_rng(6).poisson(np.exp(0.3 * eta))).astype(float), None,
                                                           dict(k=7, d=m.Poisson(), l=m.LogLink())),
                                       # (round 6) debias over the shards: the panel's all-reduce on the library's communicator (host-driven steps either way)
                                       "debias": (cols2, n2, yb, None, dict(k=6, d=m.Bernoulli(), l=m.LogitLink(), debias=True))}.items():
        p = cc.shape[0]
        lo, cnt = D.column_block(p, rank, world)
        xs = m.SnpLinAlg(cc[lo:lo + cnt], n=nn, center=True, scale=True, impute=True, device=local)
        m.profile_enable(xs, True)
        m.profile_counters(xs, reset=True)
        a = summary(D.fit_iht_sharded(yy, xs, zz, col_offset=lo, p_global=p, verbose=False, native=True, **kw))
        cnt_native = m.profile_counters(xs, reset=True)
        b = summary(D.fit_iht_sharded(yy, xs, zz, col_offset=lo, p_global=p, verbose=False, native=False, ordered_sum=True, **kw))
        cnt_callbacks = m.profile_counters(xs, reset=True)
        m.profile_enable(xs, False)
        # (round 5) with the library's own communicator the sharded fit's steps are resident on the device too (collectives queued
        # inside the gated chain); the callbacks of the host language keep the host-driven step
        out[name] = dict(native=a, callbacks=b, resident_steps_native=cnt_native["resident_steps"] + cnt_native["resident_handbacks"],
                         resident_steps_callbacks=cnt_callbacks["resident_steps"])
        if rank == 0:                                        # ... and the unsharded fit on the whole matrix
            xf = m.SnpLinAlg(cc, n=nn, center=True, scale=True, impute=True, device=local)
            out[name]["single"] = summary(m.fit_iht(yy, xf, zz, verbose=False, **kw))
            if kw.get("debias"):                             # (what the refit changes: the test asserts that it ran)
                out[name]["single_plain"] = summary(m.fit_iht(yy, xf, zz, verbose=False, **dict(kw, debias=False)))
    # the column-sharded MULTIVARIATE fit (round 5): three traits, two covariates (one of them competing in the projection)
    Ym, Zm = mv_traits(x2, np.random.default_rng(77), 3, 10, 2)
    lo, cnt = D.column_block(p2, rank, world)
    xs = m.SnpLinAlg(cols2[lo:lo + cnt], n=n2, center=True, scale=True, impute=True, device=local)
    mkw = dict(k=12, zkeep=[True, False], verbose=False)
    out["mv_r3"] = dict(native=summary_mv(D.fit_iht_sharded(Ym, xs, Zm, col_offset=lo, p_global=p2, native=True, **mkw)),
                        callbacks=summary_mv(D.fit_iht_sharded(Ym, xs, Zm, col_offset=lo, p_global=p2, native=False, ordered_sum=True, **mkw)))
    if rank == 0:
        out["mv_r3"]["single"] = summary_mv(m.fit_iht(Ym, x2, Zm, **mkw))
    del xs
    # a large model first and small ones after it: the communicator's staging buffer grows (ensure_stage) and is re-used
    lo, cnt = D.column_block(p2, rank, world)
    xs = m.SnpLinAlg(cols2[lo:lo + cnt], n=n2, center=True, scale=True, impute=True, device=local)
    ynorm = eta + rng.standard_normal(n2)
    from mendeliht_amd import dist as DD
    comm = DD.NativeComm(lo, p2, device=local)
    grow = []
    for kk in (40, 3, 90, 5):
        r1 = m.fit_iht(ynorm, xs, None, k=kk, comm=comm, verbose=False, max_iter=30)
        grow.append(dict(k=kk, logl=r1.logl, iter=int(r1.iter), nnz=int(np.count_nonzero(r1.beta))))
    comm.close()
    out["staging_growth"] = grow
    # the one exchange of a cross-validation through the library's communicator (mih_cv_allgather) against the torch all-gather
    from conftest import hash_folds
    xr = m.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True, device=local)
    folds = hash_folds(n, 3)
    cv_native = D.cv_iht_distributed(y, xr, z, native=True, path=range(1, 9), q=3, folds=folds, verbose=False)
    cv_torch = D.cv_iht_distributed(y, xr, z, native=False, path=range(1, 9), q=3, folds=folds, verbose=False)
    out["cv_gather"] = dict(native=cv_native.tolist(), torch=cv_torch.tolist())
    if rank == 0:
        out["cv_gather"]["single"] = m.cv_iht(y, xr, z, path=range(1, 9), q=3, folds=folds, verbose=False).tolist()
    dist.barrier()
    with open(out_path + f".native.r{rank}", "w") as f:
        json.dump(dict(world=world, cases=out), f)
    dist.destroy_process_group()


def main():
    out_path = sys.argv[1]
    if os.environ.get("MIH_NATIVE"):
        return native_main(out_path)
    rank, world, _ = D.init_from_env(backend="gloo")
    import torch.distributed as dist

    cases = {}

    def run(name, cols, n, y, z, **kw):
        if os.environ.get("MIH_WORKER_TRACE"):
            print("case", name, "rank", rank, file=sys.stderr, flush=True)
        p = cols.shape[0]
        lo, cnt = D.column_block(p, rank, world)
        xs = m.SnpLinAlg(cols[lo:lo + cnt], n=n, center=True, scale=True, impute=True)
        sh = D.fit_iht_sharded(y, xs, z, col_offset=lo, p_global=p, verbose=False, **kw)
        entry = dict(sharded=summary(sh), block=[lo, cnt])
        if rank == 0:
            xf = m.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)
            entry["single"] = summary(m.fit_iht(y, xf, z, verbose=False, **kw))
            if name.startswith("random") and z is not None:
                # (round 6) the single-process fit on covariates scaled by a few ulps: if the sharded fit differs, the test asks whether
                # the single-process fit reproduces ITSELF (nothing is set aside by argument)
                entry["single_nudged"] = [dict(summary(m.fit_iht(y, xf, z * g, verbose=False, **kw)), g=g) for g in (1.0 + 2.0 ** -51, 1.0 + 3 * 2.0 ** -51, 1.0 + 2.0 ** -49)]
        cases[name] = entry

    # 1. the reference's shipped example (G1): Normal, k = 7, two covariates
    n = 1000
    cols = m.read_bed(os.path.join(FIX, "normal.bed"), n)
    y = np.loadtxt(os.path.join(FIX, "normal_y_fam6.txt"))
    z = np.loadtxt(os.path.join(FIX, "covariates.txt"), delimiter=",")
    z[:, 1:] = (z[:, 1:] - z[:, 1:].mean(axis=0)) / z[:, 1:].std(axis=0, ddof=1)
    run("normal_k7", cols, n, y, z, k=7)

    # 2. logistic with missing genotypes, prior weights, a covariate that competes in the projection
    rng = np.random.default_rng(42)
    n, p = 1501, 2300                                        # ragged: not a multiple of 128 rows / 32 columns
    cols = make_bed(rng, n, p, missing_rate=0.02)
    xo = m.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)
    supp = rng.choice(p, 8, replace=False)
    eta = xo.xv_sparse(np.sort(supp), rng.standard_normal(8) * 0.7)
    zz = np.column_stack([np.ones(n), rng.standard_normal(n), rng.standard_normal(n)])
    eta = eta + 0.3 * zz[:, 1]
    yb = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    w = rng.uniform(0.5, 2.0, p)
    run("logistic_weights_zkeep", cols, n, yb, zz, k=9, d=m.Bernoulli(), l=m.LogitLink(), weight=w,
        zkeep=[True, True, False])

    # 3. Poisson, intercept only, a train mask
    lam = np.exp(0.4 * xo.xv_sparse(np.sort(supp[:4]), np.array([0.5, -0.4, 0.3, 0.6])) + 0.2)
    yp = rng.poisson(lam).astype(float)
    train = (np.arange(n) % 5 != 0).astype(np.uint8)
    run("poisson_train", cols, n, yp, None, k=6, d=m.Poisson(), l=m.LogLink(), train=train)

    # 3b. (round 6) Poisson with a planted count outlier: the working residual has one row towering over the rest, which leaves the
    #     fixed point and rides the f64 side channel of every shard's finalize (csrc/peel.h): every rank makes the same decision
    yo = yp.copy(); yo[int(np.argmin(np.abs(lam - 1.0)))] = 500.0
    run("poisson_outlier", cols, n, yo, None, k=6, d=m.Poisson(), l=m.LogLink())

    # 4. NegBin with the nuisance parameter estimated (replicated n-vector work)
    ynb = rng.negative_binomial(2, 2.0 / (2.0 + lam)).astype(float)        # overdispersed counts: r is identifiable
    run("negbin_newton", cols, n, ynb, None, k=5, d=m.NegativeBinomial(2.0), l=m.LogLink(), est_r="Newton")

    # 5. exact ties: duplicated columns straddling the shard boundary make _choose! fire globally
    lo1, cnt1 = D.column_block(p, 0, world)
    dup = cols.copy()
    if world > 1:
        dup[lo1 + cnt1] = dup[lo1 + cnt1 - 1]                # first column of rank 1 = last column of rank 0
        dup[3] = dup[lo1 + cnt1 - 1]
        dup[7] = dup[lo1 + cnt1 - 1]                         # 4 tied columns, k = 2: nonzero = 4 > k + zkeepn = 3
    run("ties_choose", dup, n, xo.xv_sparse(np.array([lo1 + cnt1 - 1]), np.array([1.0])) + 0.01 * rng.standard_normal(n),
        None, k=2)

    # 5a. init_beta (round 5: the column-sharded fit takes it): the univariate regressions are column-local, the mean of their
    #     intercepts is one scalar exchange; with a covariate that competes in the projection, and with a train mask
    run("init_beta", cols, n, eta + 0.4 + rng.standard_normal(n), zz, k=8, init_beta=True, zkeep=[True, False, True])
    run("init_beta_train", cols, n, eta + rng.standard_normal(n), None, k=6, init_beta=True, train=train)

    # 5a'. (round 6) debias: the GLM refit of the support runs on a panel summed over the shards; every shard runs the same refit.
    #      Normal with two covariates (the support settles, the refit runs from iteration 5 on), Poisson / log on the ragged matrix
    #      with missing genotypes (imputed entries in the panel), and a model small enough that a shard holds none of its columns
    run("debias_normal", cols, n, eta + 0.4 + rng.standard_normal(n), zz, k=8, debias=True)
    run("debias_poisson", cols, n, yp, None, k=6, d=m.Poisson(), l=m.LogLink(), debias=True)
    run("debias_k1", cols, n, eta + rng.standard_normal(n), None, k=1, debias=True, max_iter=20)

    # 5a". (round 6) the doubly sparse projection over the shards: groups that straddle the shard boundaries (random labels) and sorted
    #      labels, a scalar k per group (J groups of k: _choose! may fire) and a vector k (the initial projection of the gradient too)
    Gn = 12
    glab = rng.integers(1, Gn + 1, p); glab[:Gn] = np.arange(1, Gn + 1)
    run("group_random_labels", cols, n, eta + 0.3 + rng.standard_normal(n), None, group=glab, J=3, k=2)
    gs = np.sort(glab)
    run("group_sorted_labels", cols, n, yb, zz, group=gs, J=4, k=3, d=m.Bernoulli(), l=m.LogitLink())
    run("group_vector_k", cols, n, eta + rng.standard_normal(n), None, group=glab, J=3, k=rng.integers(1, 4, Gn))
    run("group_debias", cols, n, eta + 0.1 + rng.standard_normal(n), None, group=gs, J=2, k=3, debias=True)

    # 5b. seeded random cases (MIH_SWEEP_SEED for other draws): shapes down to fewer columns than one 32-column tile per rank,
    #     families, covariates in and out of zkeep, prior weights, train masks -- sharded == single, as above
    srng = np.random.default_rng(int(os.environ.get("MIH_SWEEP_SEED", 1234)))
    fams = [(m.Normal, m.IdentityLink), (m.Bernoulli, m.LogitLink), (m.Poisson, m.LogLink)]
    for t in range(4):
        rn = int(srng.integers(100, 1500)); rp = int(srng.integers(max(8, world), 700)); rq = int(srng.integers(1, 4))
        rk = int(srng.integers(1, 9)); fi = int(srng.integers(0, 3))
        rcols = make_bed(srng, rn, rp, missing_rate=float(srng.choice([0.0, 0.03])))
        xr_ = m.SnpLinAlg(rcols, n=rn, center=True, scale=True, impute=True)
        rs = np.sort(srng.choice(rp, min(4, rp), replace=False))
        rz = np.column_stack([np.ones(rn)] + [srng.standard_normal(rn) for _ in range(rq - 1)])
        reta = xr_.xv_sparse(rs, srng.standard_normal(rs.size) * 0.5) + rz @ np.concatenate([[0.3], srng.standard_normal(rq - 1) * 0.2])
        ry = [reta + srng.standard_normal(rn), (srng.random(rn) < 1 / (1 + np.exp(-reta))).astype(float),
              srng.poisson(np.exp(np.clip(0.5 * reta, -3, 3))).astype(float)][fi]
        rkw = dict(k=rk, d=fams[fi][0](), l=fams[fi][1](), max_iter=60)
        if rq > 1 and srng.random() < 0.5:
            rkw["zkeep"] = [True] + [bool(v) for v in srng.integers(0, 2, rq - 1)]
        if srng.random() < 0.4:
            rkw["weight"] = srng.uniform(0.5, 2.0, rp)
        if srng.random() < 0.4:
            rkw["train"] = (srng.random(rn) < 0.8).astype(np.uint8)
        del xr_
        run(f"random{t}", rcols, rn, ry, rz, **rkw)

    # 5c. multivariate traits (round 5: mih_fit_mv takes a column shard): the reference's shipped example, three traits with a
    #     competing covariate on the ragged matrix with missing genotypes, and tied columns across the shard boundary
    def run_mv(name, cols_, n_, Y, Z, **kw):
        if os.environ.get("MIH_WORKER_TRACE"):
            print("case", name, "rank", rank, file=sys.stderr, flush=True)
        p_ = cols_.shape[0]
        lo_, cnt_ = D.column_block(p_, rank, world)
        xs_ = m.SnpLinAlg(cols_[lo_:lo_ + cnt_], n=n_, center=True, scale=True, impute=True)
        sh_ = D.fit_iht_sharded(Y, xs_, Z, col_offset=lo_, p_global=p_, verbose=False, **kw)
        entry_ = dict(sharded=summary_mv(sh_), block=[lo_, cnt_])
        if rank == 0:
            xf_ = m.SnpLinAlg(cols_, n=n_, center=True, scale=True, impute=True)
            entry_["single"] = summary_mv(m.fit_iht(Y, xf_, Z, verbose=False, **kw))
        cases[name] = entry_

    nm = 1000
    bedm = m.read_bed(os.path.join(FIX, "multivariate.bed"), nm)
    Ymv = np.loadtxt(os.path.join(FIX, "multivariate.phen"), delimiter=",").T
    run_mv("mv_shipped", bedm, nm, Ymv, None, k=10)
    Y3, Z3 = mv_traits(xo, np.random.default_rng(78), 3, 10, 2)
    run_mv("mv_r3_cov", cols, n, Y3, Z3, k=12, zkeep=[True, False])
    # a matrix of 40 columns: with three ranks a shard's r * p entries are fewer than the K the projection asks for (it sends all it has)
    trng = np.random.default_rng(79)
    tcols = make_bed(trng, 300, 40)
    xt_ = m.SnpLinAlg(tcols, n=300, center=True, scale=True, impute=True)
    Yt_, _ = mv_traits(xt_, trng, 2, 6, 1)
    del xt_
    run_mv("mv_tiny", tcols, 300, Yt_, None, k=30, max_iter=40)
    e1 = xo.xv_sparse(np.array([lo1 + cnt1 - 1]), np.array([1.0]))
    Yt = np.vstack([e1 + 0.01 * rng.standard_normal(n), -0.5 * e1 + 0.01 * rng.standard_normal(n)])
    run_mv("mv_ties_choose", dup, n, Yt, None, k=1, max_iter=8)      # (a degenerate problem: the first steps, before rounding decides the trajectory)

    # 6. cross-validation with the (fold, k) grid sharded over the same ranks (one all-gather of the losses)
    from conftest import hash_folds
    n = 1000
    cols = m.read_bed(os.path.join(FIX, "normal.bed"), n)
    xr = m.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)        # every rank: a full replica
    folds = hash_folds(n, 3)
    mse = D.cv_iht_distributed(y, xr, z, path=range(1, 9), q=3, folds=folds, verbose=False)
    entry = dict(distributed=mse.tolist())
    if rank == 0:
        entry["single"] = m.cv_iht(y, xr, z, path=range(1, 9), q=3, folds=folds, verbose=False).tolist()
    cases["cv_grid"] = entry

    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(dict(world=world, cases=cases), f)
    else:
        with open(out_path + f".r{rank}", "w") as f:
            json.dump(dict(cases=cases), f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
