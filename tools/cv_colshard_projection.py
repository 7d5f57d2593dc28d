"""Would a COLUMN-sharded lock-step cross-validation beat the replica design on 8 GPUs?  ("design + one-GPU
projection; if the projection does not beat 0.50 s, record it and stop".)

Replica design (what mih_cv_iht does): every GPU holds all p columns and runs 12-13 of the 100 (fold, k) fits.
Column-sharded design: every GPU holds p/8 columns and advances ALL 100 fits; per round the partial X_S b_S / X_S g_S
sums and the top-k candidates of the fits in flight are exchanged.

Measured here on ONE GPU, warm (second call of each):
  a) one rank's replica share: mih_cv_iht(rank = r, world = 8) on the full matrix, r = 0..7            -> t_replica = max
  b) one rank's column share with NO exchange at all: all 100 fits on the first p/8 columns          -> t_colshard_floor
     (a LOWER bound of the column-sharded time: the per-fit n-vector work -- residuals, loglikelihoods, digit planes of up to
     36 residuals per round, k-column X v -- does not shrink with the column count, and every exchange comes on top)
and computed: what one round of the column-sharded design would put on the wire."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m

n, p, world = 500_000, 1_000_000, 8
rng = np.random.default_rng(2025)
folds = m.hash_folds(n, 5)
path = range(1, 21)
kw = dict(d=m.Bernoulli(), l=m.LogitLink(), path=path, q=5, folds=folds, verbose=False, return_raw=True)


def problem(x, pc):
    r = np.random.default_rng(2025)
    supp = np.sort(r.choice(pc, 10, replace=False))
    eta = x.xv_sparse(supp, r.standard_normal(10) * 0.5)
    return (r.random(n) < 1 / (1 + np.exp(-eta))).astype(float)


def timed(fn, reps=2):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return ts


out = {"n": n, "p": p, "world": world, "workload": "cv_iht Bernoulli/Logit path=1:20 q=5 (BASELINE configs[3])"}
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
yb = problem(x, p)
m.profile_counters(x, reset=True); m.profile_enable(x, True)
t_all = timed(lambda: m.cv_iht(yb, x, None, **kw))
cnt = m.profile_counters(x, reset=True); m.profile_enable(x, False)
out["one_gpu_all_100_fits_s"] = t_all
rounds_per_cv = cnt["rounds"] / 3.0
out["lockstep_rounds_per_cv"] = rounds_per_cv
out["replica_share_s"] = {}
for r in range(world):
    out["replica_share_s"][str(r)] = min(timed(lambda: m.cv_iht(yb, x, None, rank=r, world=world, **kw)))
t_replica = max(out["replica_share_s"].values())
del x
pc = p // world
xs = m.SnpLinAlg.synthetic(n, pc, seed=2024)
ys = problem(xs, pc)
m.profile_counters(xs, reset=True); m.profile_enable(xs, True)
t_floor = timed(lambda: m.cv_iht(ys, xs, None, **kw), reps=3)
cs = m.profile_counters(xs, reset=True); m.profile_enable(xs, False)
out["colshard_floor_all_100_fits_on_p_over_8_columns_no_exchange_s"] = t_floor
out["colshard_rounds_per_cv"] = cs["rounds"] / 4.0
# what a round would exchange: every fit in flight needs, per iteration, the sum over the shards of X_S b_S (update_xb!) and of
# X_S g_S (iht_stepsize!) -- two n-vectors of f64 -- plus ~1.3 extra update_xb! per iteration for backtracks, its top-k candidates
# (k + 1 doubles per rank, all-gather) and a few scalars; with the lanes full that is 36 fits per round pair (18 per lane)
fits_in_flight = 18
nvec_per_fit_round = 2.3
bytes_allreduce = fits_in_flight * nvec_per_fit_round * n * 8
ring_factor = 2 * (world - 1) / world                 # ring all-reduce: 2 (N-1)/N x payload over each link
xgmi_link_GBps = 153.0                                # per direction and link (MI355X_MICROARCH.md): ring collectives are per-link bound
t_wire = bytes_allreduce * ring_factor / (xgmi_link_GBps * 1e9)
out["exchange_per_lane_round"] = {
    "fits": fits_in_flight, "n_vector_allreduces_per_fit": nvec_per_fit_round, "allreduce_payload_bytes": bytes_allreduce,
    "topk_allgather_bytes": fits_in_flight * world * 21 * 8, "ring_wire_ms": 1e3 * t_wire,
    "collective_rendezvous": "at least 5 per round (step size, projection, update_xb!, convergence test; more with backtracks), each a "
                             "synchronisation of all 8 ranks and of the lane's 18 coroutines",
}
t_proj = min(t_floor) + out["colshard_rounds_per_cv"] * t_wire
out["projection_s"] = {"replica_design_max_share": t_replica, "colshard_floor_no_exchange": min(t_floor),
                       "colshard_floor_plus_wire_time": t_proj}
out["verdict"] = ("the column-sharded lock-step CV does not beat the replica design: even WITHOUT any exchange one rank's p/8 share of all 100 fits "
                  f"takes {min(t_floor):.3f} s against {t_replica:.3f} s for the slowest replica share"
                  if t_proj >= t_replica else "the column-sharded design projects faster")
print(json.dumps(out))
