// fit_lockstep.hip -- cv_iht and iht_run_many_models (src/cross_validation.jl:60-131, 232-273): the rolling lock-step drivers.  The
// (fold, k) fits of a rank advance together, ONE fused X'R pass per round scores all of them; two lanes (host thread + stream +
// workspace) pull fits from one queue.  Split out of fit.hip in round 6; one IHTVariable and its steps: fit_state.h.
#include "fit_state.h"

using namespace mih;

extern "C" {

// lock-step so that ONE pass over the 2-bit matrix serves the score of every fit in flight
// (multi-RHS X'R, up to 15 residual vectors per 5-operand pass).  Each fit keeps its own IHTVariable, backtracks and
// converges on its own; a fit that finishes is scored on its held-out samples and its slot is refilled.
struct CvFit {
    std::unique_ptr<IhtVar> v;
    std::vector<std::unique_ptr<IhtVar>> *pool = nullptr;   // the lane's free list: a finished fit hands its IHTVariable back
    void release() { if (v && pool) pool->push_back(std::move(v)); v.reset(); }
    const uint8_t *train = nullptr;  // training mask of its fold (owned by the driver, shared by the fold's fits); null: all rows
    int64_t out_index = 0;
    size_t qidx = 0;         // its number in the lanes' queue (CvQueue)
    bool fast43 = false;     // xtv_digits = -1: this round's residual rides the 43-bit format (IhtVar::residual_rides_43_bits)
    int init_key = -1;       // fits with the same key >= 0 have the same initial residual (same training rows; the model size
                             // enters only after the first score): one of them rides the pass, the others copy its X'r
    int iter = 1, nbt = 0;
    double next_logl = -std::numeric_limits<double>::infinity(), best = -std::numeric_limits<double>::infinity();
    bool done = false;
    // the lane's batched chain (k_lane_*): this fit wants its next step in the series the lane queues behind the pass / a series is in
    // flight for it / its record has been read
    bool wants_step = false, fresh_begin = false, in_batch = false, have_rec = false; ResRecord rec;
    // est_r in cv_iht: this fit is number chain_pos of chain `chain` (CvChains); when it ends it leaves its NegBin r in *chain_r
    int64_t chain = -1; size_t chain_pos = 0; double *chain_r = nullptr;
    // iht_run_many_models mode (no hold-out): where to put the finished model instead of a held-out deviance
    bool full_data = false;
    double *logl_out = nullptr; int64_t *iter_out = nullptr; double *beta_out = nullptr, *c_out = nullptr;
};

// the lane's stream waits for everything fit f has queued on its own stream (f.v->s != lane stream only with private streams)
static int fit_to_lane(CvFit &f, hipStream_t lane_s);
static int lane_to_fit(CvFit &f, hipStream_t lane_s, hipEvent_t lane_ev);

// The residuals of a lane's fits into the pass's R, the scores out of its DF, ONE launch each way instead of a copy per fit (round 6:
// 38 copies of 4 - 8 MB around every pass were 1.5 ms of the window between two passes: tools/cv_window_trace.sh)
constexpr int kLaneCopyMax = 64;
struct LaneCopy { const double *src[kLaneCopyMax]; double *dst[kLaneCopyMax]; };
__global__ void __launch_bounds__(256)
k_lane_copy(LaneCopy c, int64_t len)
{
    const double *__restrict__ a = c.src[blockIdx.y]; double *__restrict__ b = c.dst[blockIdx.y];
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int64_t n2 = len >> 1;
    if ((((uintptr_t)a | (uintptr_t)b) & 15) == 0) {
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n2; i += 256ll * gridDim.x)
            reinterpret_cast<d2 *>(b)[i] = reinterpret_cast<const d2 *>(a)[i];
        if ((len & 1) && blockIdx.x == 0 && threadIdx.x == 0) b[len - 1] = a[len - 1];
    } else
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < len; i += 256ll * gridDim.x) b[i] = a[i];
}
static void lane_copy(hipStream_t s, LaneCopy &c, int cnt, int64_t len)
{
    if (cnt <= 0) return;
    const unsigned gx = (unsigned)std::min<int64_t>((len / 2 + 255) / 256, 1024);
    hipLaunchKernelGGL(k_lane_copy, dim3(std::max(gx, 1u), (unsigned)cnt), dim3(256), 0, s, c, len);
}

static int cv_batched_xtv(const mih_mat *h, XtvWork &xw, std::vector<CvFit *> &fits, DevBuf<double> &R, DevBuf<double> &DF,
                          hipStream_t s)
{
    const int m = (int)fits.size();
    if (m == 0) return MIH_OK;
    // xtv_digits = -1: the residuals that qualified for the 43-bit format ride passes of their own (four per operand instead of
    // three); the others the 54-bit passes.  Which group a residual is in depends on itself alone.
    std::vector<CvFit *> order;
    order.reserve((size_t)m);
    for (CvFit *f : fits) if (!(xw.has_alt && f->fast43)) order.push_back(f);
    const int m54 = (int)order.size();
    for (CvFit *f : fits) if (xw.has_alt && f->fast43) order.push_back(f);
    for (int t = 0; t < m; ++t) MIH_TRY(fit_to_lane(*order[(size_t)t], s));                 // its residual is ready
    for (int t0 = 0; t0 < m; t0 += kLaneCopyMax) {
        LaneCopy c; const int cnt = std::min(m - t0, kLaneCopyMax);
        for (int t = 0; t < cnt; ++t) { c.src[t] = order[(size_t)(t0 + t)]->v->r.p; c.dst[t] = R.p + (size_t)(t0 + t) * h->n; }
        lane_copy(s, c, cnt, h->n);
    }
    if (m54) MIH_TRY(xtv_device(h, xw, R.p, m54, DF.p, s));
    if (m > m54) {
        xw.use_alt = true;
        const int rc = xtv_device(h, xw, R.p + (size_t)m54 * h->n, m - m54, DF.p + (size_t)m54 * h->p, s);
        xw.use_alt = false;
        MIH_TRY(rc);
        h->prof->count(MIH_CNT_RESIDUALS_43BIT, m - m54);
    }
    for (int t0 = 0; t0 < m; t0 += kLaneCopyMax) {
        LaneCopy c; const int cnt = std::min(m - t0, kLaneCopyMax);
        for (int t = 0; t < cnt; ++t) { c.src[t] = DF.p + (size_t)(t0 + t) * h->p; c.dst[t] = order[(size_t)(t0 + t)]->v->df.p; }
        lane_copy(s, c, cnt, h->p);
    }
    MIH_HIP(hipGetLastError());
    return MIH_OK;
}

static int cv_finish(CvFit &f, double *mses_raw)
{
    if (probe_env("MENDELIHT_CV_TRACE"))             // measurement build: which fit took how many iterations (the queue's order is built on it)
        fprintf(stderr, "fit out_index %lld k %lld: %d iterations\n", (long long)f.out_index, (long long)f.v->k, f.iter);
    if (f.v->res_active) MIH_TRY(f.v->res_end(&f.next_logl, &f.best));          // (a resident fit's iterate and its two loglikelihoods come home)
    f.best = f.v->save_prev(f.next_logl, f.best);
    MIH_TRY(f.v->save_best_model());
    if (f.chain_r) *f.chain_r = f.v->nb_r;             // v.d stays as the last mle_for_r left it (cross_validation.jl:91,110)
    if (f.full_data) {                                 // iht_run_many_models: the fitted model itself is the result
        if (f.logl_out) *f.logl_out = f.best;
        if (f.iter_out) *f.iter_out = f.iter;
        if (f.beta_out) {
            std::memset(f.beta_out, 0, sizeof(double) * (size_t)f.v->p);
            for (size_t t = 0; t < f.v->best_b.idx.size(); ++t) f.beta_out[f.v->best_b.idx[t]] = f.v->best_b.val[t];
        }
        if (f.c_out) for (int l = 0; l < f.v->q; ++l) f.c_out[l] = f.v->best_c[l];
        f.done = true;
        f.v->h->prof->count(MIH_CNT_FITS, 1);
        f.release();
        return MIH_OK;
    }
    MIH_TRY(f.v->set_weights(f.train, 1));            // cv_wts <- test mask (cross_validation.jl:115-116)
    MIH_TRY(f.v->update_xb());                        // predict! (:279-286)
    double dev;
    MIH_TRY(f.v->mu_loglik(1, nullptr, &dev));
    mses_raw[f.out_index] = dev;
    f.done = true;
    f.v->h->prof->count(MIH_CNT_FITS, 1);
    f.release();                                       // the IHTVariable (device buffers, column cache) goes back to the lane's pool
    return MIH_OK;
}

static int fit_to_lane(CvFit &f, hipStream_t lane_s)
{
    if (f.v->s == lane_s || !f.v->ev) return MIH_OK;
    MIH_HIP(hipEventRecord(f.v->ev, f.v->s));
    MIH_HIP(hipStreamWaitEvent(lane_s, f.v->ev, 0));
    return MIH_OK;
}
static int lane_to_fit(CvFit &f, hipStream_t lane_s, hipEvent_t lane_ev)        // lane_ev has been recorded on lane_s
{
    if (f.v->s == lane_s || !f.v->ev) return MIH_OK;
    MIH_HIP(hipStreamWaitEvent(f.v->s, lane_ev, 0));
    return MIH_OK;
}

// The rolling lock-step driver.  `cap` slots; in every round each occupied slot needs exactly one score pass -- a
// fit that has just been created its initial score (init_pre / init_post, utilities.jl:366-438), a running fit the
// score that ends its step (step_pre / step_post) -- so ONE fused pass serves all of them, and the slot of a fit
// that finished is refilled from the queue in the next round: the passes stay full until the queue is empty.
// make(i, f) sets up fit number i (its IhtVar, training mask, output slots).
// The lane keeps the IHTVariables of finished fits and hands them to the fits it starts next (the reference re-uses one
// IHTVariable per thread the same way, cross_validation.jl:91,110): ~25 hipMalloc / hipFree per fit otherwise, and every
// hipFree waits for the OTHER lane's fused pass to finish.
struct CvShared {                 // what a lane shares with its fits
    double *y = nullptr, *z = nullptr;       // the lane's device copies of y and z (read-only)
    IbShared *ib = nullptr;                  // the lane's cache of the initialize_beta! regressions (init_beta = true)
    std::vector<hipStream_t> streams;        // non-empty: the fits queue their small kernels on these, round-robin (LaneSched)
    mutable size_t rr = 0;
    hipStream_t next_stream() const { return streams.empty() ? nullptr : streams[rr++ % streams.size()]; }
};
using MakeFit = std::function<int(size_t, CvFit &, hipStream_t, const CvShared &)>;

// est_r != :None in cv_iht.  The reference builds ONE IHTVariable per Julia thread and re-uses it for every (fold, k) combination
// the thread is given (cross_validation.jl:91,103,110); init_iht_indices! resets everything but v.d, so the NegBin r that
// mle_for_r left at the end of one fit is the starting value of the thread's next fit.  `Threads.@threads :static` gives thread t
// a contiguous block of the fold-major combinations, so the fits form one CHAIN per thread: the chains are independent of each
// other and advance in lock-step (the queue hands out chains instead of fits; a slot that finishes a fit starts the next fit of
// its chain), the fits of a chain run one after the other.
struct CvChains {
    std::vector<std::vector<size_t>> fits;       // fits[c]: the fit numbers (arguments of make) of chain c in the thread's order
    std::vector<double> r;                       // r[c]: what the chain's next fit starts from
};

// The queue the lanes draw their fits from.  Plain order (the caller's: fold-major) until something is known; then longest first:
// the fits of a cross-validation that share a model size k take nearly the same number of iterations in every fold (5 .. 17 at
// configs[3], the same to within one or two across the folds), so once a fit of some k has finished, the remaining fits of that
// k have a forecast -- and a k none of whose fits has finished yet is one whose first fit is STILL RUNNING: the longest kind.
// Unknown first, then by descending forecast, ties in the caller's order.  The order changes which fits share a pass, never a
// result (every fit is independent of its company: DESIGN.md 3.4).  The longest fits then start early and the tail, where the
// passes run half empty, is short.
struct CvQueue {
    std::mutex mu;
    size_t total = 0, ntaken = 0;
    std::vector<char> taken;
    std::vector<int> key;            // key[i]: what fit i shares its length with (the index of its k in the path); empty: plain order
    std::vector<int> seen;           // seen[key]: most iterations a finished fit of that key took (0: none finished)
    bool stop = false;               // an error somewhere: hand out nothing more
    long unknown_bias = 0;
    void init(size_t n, std::vector<int> keys)
    {
        if (const char *e = probe_env("MENDELIHT_CV_ORDER")) unknown_bias = !strcmp(e, "kdesc") ? 1 : !strcmp(e, "kasc") ? -1 : 0;
        total = n; ntaken = 0; taken.assign(n, 0); key = std::move(keys);
        int kmax = -1; for (int v : key) kmax = std::max(kmax, v);
        seen.assign((size_t)(kmax + 1), 0);
    }
    bool pick(size_t *out)
    {
        std::lock_guard<std::mutex> g(mu);
        if (stop || ntaken >= total) return false;
        size_t best = total; long bestp = -1;
        for (size_t i = 0; i < total; ++i) {
            if (taken[i]) continue;
            // (unknown lengths: in the caller's order, or -- measurement build, MENDELIHT_CV_ORDER=kdesc / kasc -- the larger / smaller model sizes first)
            const long pr = key.empty() ? 0 : (seen[(size_t)key[i]] == 0 ? (1l << 30) + unknown_bias * (long)key[i] : (long)seen[(size_t)key[i]]);
            if (pr > bestp) { bestp = pr; best = i; }
            if (key.empty()) break;
        }
        taken[best] = 1; ++ntaken;
        *out = best;
        return true;
    }
    void report(size_t i, int iterations)
    {
        if (key.empty()) return;
        std::lock_guard<std::mutex> g(mu);
        int &sv = seen[(size_t)key[i]];
        sv = std::max(sv, iterations);
    }
    void halt() { std::lock_guard<std::mutex> g(mu); stop = true; }
};

// Tail of the queue: once no new fits are left, the fits of both lanes thin out and two half-empty fused passes cost far more
// than one fuller pass (6 + 6 residuals: 2 x 20.9 ms, 12 in one pass: 31.6 ms).  Lane 1 therefore hands ALL its fits over to
// lane 0 as soon as they fit into lane 0's free slots, and ends.  A fit is handed over between two rounds, when everything it
// queued on its lane's stream has completed; the adopting lane re-points it to its own stream and pool.
struct CvHandover {
    std::mutex mu;
    std::vector<std::unique_ptr<CvFit>> orphans;    // handed over by lane 1, not yet adopted by lane 0
    bool accepting = true;                          // lane 0 is still running rounds
    std::atomic<int> active0{1 << 30};              // occupied slots of lane 0 (published once its view of the queue is drained)
};

static int cv_run_rolling(const mih_mat *h, const mih_fit_params &pr, size_t total, CvQueue &queue, int cap,
                          const MakeFit &make, XtvWork &xw, DevBuf<double> &R,
                          DevBuf<double> &DF /* (cap + init_slots) x p */, hipStream_t s, double *mses_raw, CvHandover *ho = nullptr, int lane_id = 0,
                          int init_slots = 0, const CvShared &shared = CvShared(), std::atomic<int> *inflight = nullptr,
                          CvChains *chains = nullptr /* total = number of chains */)
{
    std::vector<int64_t> cont_chain((size_t)cap, -1);     // the chain a free slot goes on with, and the position of its next fit
    std::vector<size_t> cont_pos((size_t)cap, 0);
    std::vector<std::unique_ptr<IhtVar>> pool;            // declared before the slots: outlives them
    std::vector<std::unique_ptr<CvFit>> slot((size_t)cap);
    std::vector<CvFit *> need, riders;
    std::vector<char> fresh;
    std::map<int, double *> df0;                                         // initial X'r per init_key: slots behind the pass's outputs in DF
    std::vector<std::pair<CvFit *, double *>> owners, followers;
    static const bool share_init = probe_env("MENDELIHT_CV_NO_INIT_SHARE") == nullptr;
    bool drained = false;                 // the shared queue is empty
    LaneSched sched;
    sched.enabled = !shared.streams.empty();
    hipEvent_t lane_ev = nullptr;         // "the lane's stream has got this far": the fits' streams wait for it behind the pass
    MIH_HIP(hipEventCreateWithFlags(&lane_ev, hipEventDisableTiming));
    struct EvGuard { hipEvent_t e; ~EvGuard() { (void)hipEventDestroy(e); } } ev_guard{lane_ev};
    // ---- the lane's batched chain (round 6): the records its kernels read, and the two halves of a round's step --------------------
    // Which way a lane's resident fits step (tools/ab_cv_lanes.sh, tools/ab_cv_share.sh; configs[3], same box):
    //   one chain per fit on the fit's own stream (lane_queue_step)   2.43-2.46 s for the 100 fits, 0.42-0.44 s for a rank's 13   <- the library
    //   ONE batched chain per lane round (launch_series, k_lane_*)    2.58-2.63 s, 0.43-0.45 s     (measurement build: MENDELIHT_LANE_BATCHED=1)
    //   host-driven steps (step_mode 1, rounds 1-5)                   2.42-2.49 s, 0.42-0.46 s
    // The small kernels of a round can only run in the window between two fused passes (a pass's workgroups hold every CU), so what
    // counts is their GPU time, not the host's waits: the batched chain needs 30 launches per round instead of ~250, but its two
    // wide products (k_lane_xb, k_lane_xgk: 0.75 and 0.83 ms for 19 fits, 0.7 TB/s) and the attempt slots it queues for every fit
    // whether it backtracks or not cost more than the launches saved.  Kept, bit for bit with the others, as the base for that work.
    static const bool per_fit_chains = probe_env("MENDELIHT_LANE_BATCHED") == nullptr;
    PinBuf<LaneFit> largs_h; DevBuf<LaneFit> largs_d;
    { ArenaScope own_buffers(nullptr); MIH_TRY(largs_h.alloc((size_t)cap * 2, true)); MIH_TRY(largs_d.alloc((size_t)cap * 2)); }
    struct DrainFirst { hipStream_t s; ~DrainFirst() { (void)hipStreamSynchronize(s); } } drain_first{s};      // (an error return: no kernel may still read the records above)
    int largs_slot = 0;
    std::vector<CvFit *> in_flight;                          // fits with a series of the batched chain queued and its record not read yet
    // one series for `fits` on the lane's stream: (step_start) Z'r, df on the support, X_S df_S, the step size; then ONE attempt
    auto launch_series = [&](std::vector<CvFit *> &fits, bool step_start) -> int {
        const int F = (int)fits.size();
        if (F == 0) return MIH_OK;
        LaneFit *hp = largs_h.p + (size_t)largs_slot * cap; LaneFit *dp = largs_d.p + (size_t)largs_slot * cap;
        largs_slot ^= 1;
        const IhtVar &v0 = *fits[0]->v;
        int64_t kc = 0; bool any_score = false;
        for (int t = 0; t < F; ++t) {
            IhtVar &v = *fits[(size_t)t]->v;
            v.lane_seq = ++v.res_seq;
            const bool new_score = step_start && !fits[(size_t)t]->fresh_begin;
            v.lane_fill(hp[t], v.lane_seq, step_start, new_score);
            any_score = any_score || new_score;
            kc = std::max(kc, v.res_kcap);
            fits[(size_t)t]->in_batch = true; fits[(size_t)t]->have_rec = false;
        }
        MIH_HIP(hipMemcpyAsync(dp, hp, sizeof(LaneFit) * (size_t)F, hipMemcpyHostToDevice, s));
        const bool fix = v0.res_fix();
        const unsigned wide = v0.res_wide_blocks(), nbk = (unsigned)v0.nb;
        if (step_start) {
            if (any_score) {
                hipLaunchKernelGGL(k_lane_zt_r, dim3(kZtrBlocks, (unsigned)v0.q, (unsigned)F), dim3(256), 0, s, dp);
                hipLaunchKernelGGL(k_lane_support, dim3((unsigned)nblk(kc), (unsigned)F), dim3(256), 0, s, dp);
            }
            if (!fix) hipLaunchKernelGGL(k_lane_xgk<false>, dim3(wide, (unsigned)F), dim3(kResWideRows), kResWideLds, s, dp);
            else {
                hipLaunchKernelGGL(k_lane_xgk<true>, dim3(wide, (unsigned)F), dim3(kResWideRows), kResWideLds, s, dp);
                hipLaunchKernelGGL(k_lane_missing, dim3(1, (unsigned)F), dim3(1024), 0, s, dp, -1, 0);
                hipLaunchKernelGGL(k_lane_stepsize, dim3(nbk, (unsigned)F), dim3(256), 0, s, dp);
            }
            hipLaunchKernelGGL(k_lane_eta, dim3(1, (unsigned)F), dim3(256), 0, s, dp);
        }
        // attempt slots: each serves whichever attempt a fit is due (its control block counts); a fit whose step has stood finds its
        // gate closed in the slots behind.  The logistic fits of configs[3] backtrack about once per step, and a slot that turns out
        // empty for every fit costs six launches that exit at once -- less than the round trip a series cut short would cost all
        // of its fits: 1 + max_step slots, so a series always ends every step it began (attempt max_step stands: utilities.jl:484)
        const int slots = 1 + std::max(0, pr.max_step);
        for (int j = 0; j < slots; ++j) {
            hipLaunchKernelGGL(k_lane_grad, dim3(kResGradBlocks, (unsigned)F), dim3(256), 0, s, dp);
            hipLaunchKernelGGL(k_lane_hist2, dim3(kResHistBlocks, (unsigned)F), dim3(256), 0, s, dp);
            hipLaunchKernelGGL(k_lane_collect, dim3(kResCollectBlocks, (unsigned)F), dim3(256), 0, s, dp);
            hipLaunchKernelGGL(k_lane_select, dim3(1, (unsigned)F), dim3(1024), 0, s, dp);
            if (!fix) hipLaunchKernelGGL(k_lane_xb<false>, dim3(wide, (unsigned)F), dim3(kResWideRows), kResWideLds, s, dp);
            else {
                hipLaunchKernelGGL(k_lane_xb<true>, dim3(wide, (unsigned)F), dim3(kResWideRows), kResWideLds, s, dp);
                hipLaunchKernelGGL(k_lane_missing, dim3(1, (unsigned)F), dim3(1024), 0, s, dp, 0, 1);
                hipLaunchKernelGGL(k_lane_mu, dim3(nbk, (unsigned)F), dim3(256), 0, s, dp);
            }
            hipLaunchKernelGGL(k_lane_decide, dim3(1, (unsigned)F), dim3(256), 0, s, dp, j + 1 < slots ? 1 : 0);
        }
        MIH_HIP(hipGetLastError());
        return MIH_OK;
    };
    // the records of the series in flight; a fit whose step is still backtracking rides another series (its attempt counter says which
    // attempt is due), until every fit's step has stood, stopped or been handed back
    auto collect_series = [&]() -> int {
        std::vector<CvFit *> again;
        while (!in_flight.empty()) {
            again.clear();
            for (CvFit *f : in_flight) {
                MIH_TRY(f->v->res_wait(f->v->lane_seq, &f->rec, s));
                if (f->rec.status == RES_PENDING) { f->v->h->prof->count(MIH_CNT_RESIDENT_ATTEMPTS, 1); again.push_back(f); }
                else { f->have_rec = true; f->in_batch = false; }
            }
            for (CvFit *f : again) f->fresh_begin = false;
            MIH_TRY(launch_series(again, false));
            in_flight = again;
        }
        return MIH_OK;
    };
    struct SlotOut { CvFit *f = nullptr; char fresh = 0; };
    std::vector<SlotOut> outs((size_t)cap);
    std::vector<std::function<int()>> tasks;
    auto occupied = [&]() { int c = 0; for (auto &sl : slot) c += sl != nullptr; return c; };
    auto adopt = [&]() {                  // lane 0: take handed-over fits into free slots (caller holds ho->mu)
        for (int t = 0; t < cap && !ho->orphans.empty(); ++t)
            if (!slot[t]) {
                slot[t] = std::move(ho->orphans.back());
                ho->orphans.pop_back();
                slot[t]->pool = &pool;
                if (!slot[t]->v->ev) slot[t]->v->s = s;            // it ran on lane 1's own stream (worker streams belong to the matrix)
            }
    };
    static const bool trace_rounds = probe_env("MENDELIHT_CV_TRACE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_round = tnow(), t_pre = 0.0, t_post = 0.0;
    int round_no = 0;
    // everything of a round that one slot does BEFORE the fused pass: finish or refill, then the step up to its residual
    auto slot_pre = [&](int t) -> int {
        for (;;) {
            if (!slot[t]) {
                size_t i; int64_t ch = -1;
                if (chains && cont_chain[(size_t)t] >= 0) { ch = cont_chain[(size_t)t]; i = chains->fits[(size_t)ch][cont_pos[(size_t)t]]; }
                else {
                    if (drained) return MIH_OK;
                    if (!queue.pick(&i)) { drained = true; return MIH_OK; }
                    if (chains) { ch = (int64_t)i; cont_pos[(size_t)t] = 0; i = chains->fits[(size_t)ch][0]; }
                }
                slot[t].reset(new CvFit());
                slot[t]->pool = &pool;
                if (!pool.empty()) { slot[t]->v = std::move(pool.back()); pool.pop_back(); }
                MIH_TRY(make(i, *slot[t], s, shared));
                CvFit &f = *slot[t];
                f.qidx = i;
                if (chains) {
                    f.chain = ch; f.chain_pos = cont_pos[(size_t)t]; f.chain_r = &chains->r[(size_t)ch];
                    f.v->nb_r = chains->r[(size_t)ch];
                    cont_chain[(size_t)t] = -1;
                }
                MIH_TRY(f.v->init_pre(f.train));
                if (share_init && f.init_key >= 0) {
                    auto it = df0.find(f.init_key);
                    if (it != df0.end()) {       // its initial X'r is known from an earlier round: no pass, straight on to its first step
                        MIH_HIP(hipMemcpyAsync(f.v->df.p, it->second, sizeof(double) * h->p, hipMemcpyDeviceToDevice, f.v->s));    // (written on the lane's stream rounds ago)
                        MIH_TRY(f.v->init_post());
                        if (f.iter < pr.max_iter) MIH_TRY(f.v->lane_queue_step(f.next_logl, f.best, f.iter - 1, &pr));    // (its first step, resident)
                        h->prof->count(MIH_CNT_SHARED_INIT, 1);
                        h->prof->count(MIH_CNT_INIT_SCORES, 1);     // its initial score, served by a copy
                        continue;
                    }
                }
                if (f.v->auto_digits()) MIH_TRY(f.v->residual_rides_43_bits(&f.fast43));      // (the initial residual: y - mu of the intercept)
                outs[(size_t)t] = SlotOut{&f, 1};
                return MIH_OK;
            }
            CvFit &f = *slot[t];
            if (!f.done && f.iter >= pr.max_iter) MIH_TRY(cv_finish(f, mses_raw));        // fit.jl:170-179
            if (f.done) {                                                                 // refill this slot
                if (!chains) queue.report(f.qidx, f.iter);                                // (how long fits of its model size take: CvQueue)
                if (chains && f.chain >= 0 && f.chain_pos + 1 < chains->fits[(size_t)f.chain].size()) {
                    cont_chain[(size_t)t] = f.chain; cont_pos[(size_t)t] = f.chain_pos + 1;      // ... with the next fit of its chain
                }
                slot[t].reset(); continue;
            }
            // (round 6) a resident fit: the step's chain was queued behind the last pass (slot_post); its record is read here
            bool stepped = false, dev_stop = false; double dev_tol = 0.0;
            if (f.have_rec) {                                   // (the lane's batched chain: collect_series has read this fit's record)
                bool again = false;
                f.have_rec = false;
                MIH_TRY(f.v->lane_take_record(f.rec, &f.next_logl, &f.best, &f.nbt, &dev_tol, &stepped, &again));
            } else
            MIH_TRY(f.v->lane_collect_step(&pr, &f.next_logl, &f.best, &f.nbt, &dev_tol, &stepped, &dev_stop));
            if (!stepped) {
                f.best = f.v->save_prev(f.next_logl, f.best);
                MIH_TRY(f.v->step_pre(f.next_logl, pr.max_step, &f.nbt, &f.next_logl));
            }
            h->prof->count(MIH_CNT_SCORES, 1);                                            // an IHT iteration (fit.jl's counter)
            // (round 5) debias! (fit.jl:188) and the convergence test (fit.jl:197) need nothing of the score that ends this step:
            // they look at b, b0, c, c0 only.  A fit that converges HERE is finished (save_best_model, predict!) without riding
            // the pass -- the reference computes that last score inside iht_one_step! and never reads it -- and its slot is
            // refilled in this same round: one residual in a hundred fewer per fit, 100 of 1157 at configs[3].
            IhtVar &v = *f.v;
            if (v.debias && f.iter >= 5 && v.b.idx == v.b0.idx && !v.b.idx.empty())          // fit.jl:188: v.idx == v.idx0 && debias!(v)
                MIH_TRY(debias_glm_device(h, v.b.idx.data(), (int64_t)v.b.idx.size(), v.y.p, v.dist, v.link, v.nb_r, v.b.val.data(), v.s));
            const double sc = stepped ? dev_tol : v.check_convergence();                   // (k_res_select's tol: the same maxima)
            if (f.iter >= pr.min_iter && sc < pr.tol) {
                if (std::isnan(f.next_logl)) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
                if (std::isinf(f.next_logl)) { set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL; }
                h->prof->count(MIH_CNT_SKIPPED_LAST_SCORES, 1);
                MIH_TRY(cv_finish(f, mses_raw));                                          // fit.jl:197-203
                continue;                                                                 // (f.done: the slot is refilled above)
            }
            f.fast43 = false;
            if (v.auto_digits()) MIH_TRY(v.residual_rides_43_bits(&f.fast43));
            outs[(size_t)t] = SlotOut{&f, 0};
            return MIH_OK;
        }
    };
    // ... and AFTER it: the fit takes its X'r, ends the step (or its initialisation) and decides whether it is done
    auto slot_post = [&](CvFit *f, char is_fresh) -> int {
        if (is_fresh) MIH_TRY(f->v->init_post());
        else {
            if (!f->v->res_active) MIH_TRY(f->v->step_post(f->next_logl));      // (debias! and the convergence test of this step ran before the pass: slot_pre)
            f->iter++;
        }
        // the next step goes out now, behind the pass, without a wait (a fit that does not qualify steps host-driven in slot_pre):
        // as one of the lane's batched series (launch_series, behind these tasks), or as a chain of its own on its stream
        f->wants_step = false; f->fresh_begin = false;
        if (f->iter >= pr.max_iter || !f->v->res_ok) return MIH_OK;
        if (per_fit_chains) return f->v->lane_queue_step(f->next_logl, f->best, f->iter - 1, &pr);
        if (!f->v->res_active) {
            if (f->v->res_begin(f->next_logl, f->best, f->iter - 1, 1, &pr) != MIH_OK) { f->v->res_ok = false; return MIH_OK; }
            f->fresh_begin = true;
        }
        f->wants_step = true;
        return MIH_OK;
    };
    for (;;) {
        h->prof->count(MIH_CNT_ROUNDS, 1);
        MIH_TRY(collect_series());                       // (every fit of this lane is quiescent again: what the hand-over below relies on)
        if (trace_rounds) {
            const double t = tnow();
            fprintf(stderr, "lane %d round %d: %.2f ms (before the pass %.2f ms, behind it %.2f ms of host time), %zu scores\n", lane_id, round_no++,
                    t - t_round, t_pre, t_post, need.size());
            t_round = t;
        }
        if (ho && lane_id == 0) {
            std::lock_guard<std::mutex> g(ho->mu);
            adopt();
            if (drained) ho->active0.store(occupied() + (int)ho->orphans.size());
        }
        if (ho && lane_id == 1 && drained) {             // between two rounds: every fit of this lane is quiescent
            for (auto &sl : slot) if (sl && sl->done) sl.reset();         // finished in the last round: nothing to hand over
            const int mine = occupied();
            std::lock_guard<std::mutex> g(ho->mu);
            if (mine > 0 && ho->accepting && mine + ho->active0.load() <= cap) {
                //  a handed-over fit is quiescent: if it ran on this lane's stream it forgets it -- the stream is destroyed
                // when the lane returns, and an orphan that is never adopted (lane 0 failed) must not synchronise a dead stream
                //  ... and the lane's cache of the initialize_beta! regressions, which lives on this lane's stack (the fit
                // is past init_beta_phase; make() re-points a recycled IHTVariable)
                for (auto &sl : slot) if (sl) { sl->pool = nullptr; if (!sl->v->ev) sl->v->s = nullptr; sl->v->ib_shared = nullptr; sl->v->ib_key = -1; ho->orphans.push_back(std::move(sl)); }
                ho->active0.fetch_add(mine);
                h->prof->count(MIH_CNT_HANDOVERS, 1);
                if (inflight) inflight[lane_id].store(0);
                return MIH_OK;                           // lane 0 finishes them
            }
        }
        need.clear(); fresh.clear();
        const double t_a = tnow();
        tasks.clear();
        for (int t = 0; t < cap; ++t) { outs[(size_t)t] = SlotOut(); tasks.emplace_back([&slot_pre, t]() { return slot_pre(t); }); }
        MIH_TRY(sched.run(tasks));
        for (int t = 0; t < cap; ++t) if (outs[(size_t)t].f) { need.push_back(outs[(size_t)t].f); fresh.push_back(outs[(size_t)t].fresh); }
        t_pre = tnow() - t_a;
        if (need.empty()) {
            if (ho && lane_id == 0) {                    // leave only when nothing was handed over in the meantime
                std::lock_guard<std::mutex> g(ho->mu);
                if (!ho->orphans.empty()) continue;
                ho->accepting = false;
            }
            break;
        }
        // Initial scores are shared: init_iht_indices! (utilities.jl:366-438) computes its first X'r from b = 0 and the intercept
        // of the training rows before the model size k plays any role, so the 20 fits of a fold (or all fits of a model path)
        // start from the SAME residual.  One fit per key rides the pass; its X'r is kept for the fits of that key this lane
        // starts later (100 -> at most 10 initial scores per cross-validation, 7 % of all scores).
        riders.clear(); followers.clear();
        for (size_t t = 0; t < need.size(); ++t) {
            CvFit *f = need[t];
            const int key = (fresh[t] && share_init) ? f->init_key : -1;
            if (key < 0) { riders.push_back(f); continue; }
            auto it = df0.find(key);
            if (it != df0.end()) { followers.emplace_back(f, it->second); h->prof->count(MIH_CNT_SHARED_INIT, 1); continue; }
            riders.push_back(f);
            if ((int)df0.size() < init_slots) {            // room in the lane's cache (allocated with its workspace)
                double *buf = DF.p + ((size_t)cap + df0.size()) * (size_t)h->p;
                df0[key] = buf;
                owners.emplace_back(f, buf);
            }
        }
        if (h->prof->on) {
            const int mine_now = occupied();
            h->prof->count_max(MIH_CNT_MAX_LANE_SLOTS, mine_now);
            if (inflight) { inflight[lane_id].store(mine_now); h->prof->count_max(MIH_CNT_MAX_IN_FLIGHT, inflight[0].load() + inflight[1].load()); }
            else h->prof->count_max(MIH_CNT_MAX_IN_FLIGHT, mine_now);
            //  an IHT iteration is a STEP's score (fit.jl counts no iteration for init_iht_indices!'s score): the initial
            // scores -- most of them served by a copy -- are counted on their own
            int64_t nfresh = 0;
            for (char fr : fresh) nfresh += fr != 0;
            h->prof->count(MIH_CNT_INIT_SCORES, nfresh);
        }
        MIH_TRY(cv_batched_xtv(h, xw, riders, R, DF, s));
        for (auto &o : owners) MIH_HIP(hipMemcpyAsync(o.second, o.first->v->df.p, sizeof(double) * h->p, hipMemcpyDeviceToDevice, s));
        owners.clear();
        for (auto &fo : followers) {
            MIH_TRY(fit_to_lane(*fo.first, s));            // (its stream has nothing pending on df, but keep the order explicit)
            MIH_HIP(hipMemcpyAsync(fo.first->v->df.p, fo.second, sizeof(double) * h->p, hipMemcpyDeviceToDevice, s));
        }
        MIH_HIP(hipEventRecord(lane_ev, s));
        for (CvFit *f : need) MIH_TRY(lane_to_fit(*f, s, lane_ev));
        const double t_b = tnow();
        tasks.clear();
        for (size_t t = 0; t < need.size(); ++t) { CvFit *f = need[t]; const char fr = fresh[t]; tasks.emplace_back([&slot_post, f, fr]() { return slot_post(f, fr); }); }
        MIH_TRY(sched.run(tasks));
        {
            std::vector<CvFit *> wants;
            for (CvFit *f : need) if (f->wants_step) {
                f->wants_step = false;
                if (f->fresh_begin) MIH_TRY(fit_to_lane(*f, s));          // (res_begin's uploads went through the fit's own stream)
                wants.push_back(f);
            }
            MIH_TRY(launch_series(wants, true));
            for (CvFit *f : wants) { f->fresh_begin = false; in_flight.push_back(f); }
        }
        t_post = tnow() - t_b;
    }
    return MIH_OK;
}

// A lane's stream, on which its fused passes run.  (round 6) The passes leave a few CUs of the chip alone (a CU mask on the stream):
// a fused pass's workgroups hold every CU they can get until the kernel's last wave of workgroups, so the small per-fit kernels of
// BOTH lanes could only run in the window between two passes (228 ms of configs[3]'s 2.43 s); with `reserve` CUs kept
// out of the passes' reach they run WHILE the other lane's pass is in flight.  The pass is bound by the matrix pipe under the
// package power cap, not by the number of CUs: what it loses in CUs it gets back in clock (measured: tools/ab_cv_lanes.sh).
static int lane_cu_reserve()
{
    static const int v = [] { const char *e = probe_env("MENDELIHT_LANE_CU_RESERVE"); return e ? atoi(e) : kLaneCuReserve; }();
    return v;
}
static int lane_stream_create(const mih_mat *h, hipStream_t *out)
{
    const int reserve = lane_cu_reserve();
    if (reserve > 0) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, h->device) == hipSuccess && pr.multiProcessorCount > 2 * reserve) {
            const int cus = pr.multiProcessorCount, words = (cus + 31) / 32;
            std::vector<uint32_t> mask((size_t)words, 0xffffffffu);
            if (cus & 31) mask[(size_t)words - 1] = (1u << (cus & 31)) - 1u;
            // Bit b of the mask is CU b / 8 of XCD b % 8 (tools/cu_mask_map.hip, profiles/r06_cu_mask_map.json), and a kernel's
            // workgroups are dealt to the XCDs round-robin whatever CUs each has: the reserved CUs must come EVENLY from the XCDs, or
            // the XCD that lost most is the straggler of every pass (the round's first attempt took every 32nd bit -- all of
            // them CUs of XCD 7: 8 reserved CUs cost 25 %, 16 cost 66 %, and 32, a whole XCD, were not honoured at all).
            const int xcds = (cus % 8 == 0 && cus >= 64) ? 8 : 1, per_xcd = cus / xcds, take = std::max(1, reserve / xcds);
            for (int x = 0; x < xcds; ++x)
                for (int t = 0; t < take; ++t) {
                    const int b = (per_xcd - 1 - t * (per_xcd / take)) * xcds + x;
                    mask[(size_t)(b >> 5)] &= ~(1u << (b & 31));
                }
            if (hipExtStreamCreateWithCUMask(out, (uint32_t)words, mask.data()) == hipSuccess) return MIH_OK;
            (void)hipGetLastError();
        } else (void)hipGetLastError();
    }
    MIH_HIP(hipStreamCreate(out));
    return MIH_OK;
}

// Two rolling drivers ("lanes"), each with its own host thread, stream and fused-pass workspace, pull fits from one
// queue: while one lane's host thread walks the small per-fit kernel chains between two passes (about 0.5 ms per
// fit and round), the other lane's fused pass keeps the GPU busy.  Every fit is independent of the lane it runs in.
static int cv_run_lanes(const mih_mat *h, const mih_fit_params &pr, size_t total, const MakeFit &make, double *mses_raw, int init_keys,
                        const double *y_host, const double *z_host, int64_t q, CvChains *chains = nullptr,
                        const std::vector<int> &queue_keys = std::vector<int>() /* per fit: what it shares its length with (CvQueue) */)
{
    if (chains) total = chains->fits.size();               // the queue hands out chains
    const XtvTune tune = xtv_tune(&pr);
    if (!xtv_digits_valid(tune.digits)) { set_error("residual format must be 0 (default), -1 (auto in lock-step drivers), 4910, 4908, 1316, 1308 or 428"); return MIH_BAD_ARG; }
    const int width = xtv_lockstep_width(h, tune);
    int lanes = total > (size_t)width / 2 ? 2 : 1;  // more fits than one full pass holds: two lanes hide each other's per-fit chains (25 fits: 1.01 s against 1.06 s with one lane; 13 fits: 0.56 s with one lane, 0.65 s with two)
    if (const char *e = probe_env("MENDELIHT_CV_LANES")) { int v = atoi(e); if (v >= 1 && v <= 4) lanes = (int)std::min<size_t>((size_t)v, total); }
    const int cap = (int)std::min<size_t>((size_t)std::max(1, width / lanes), (total + lanes - 1) / lanes);
    CvQueue queue;
    queue.init(total, (chains || probe_env("MENDELIHT_CV_PLAIN_ORDER")) ? std::vector<int>() : queue_keys);
    CvHandover handover;
    std::atomic<int> inflight[2];
    inflight[0].store(0); inflight[1].store(0);
    const bool merge_tail = lanes == 2 && !chains && !probe_env("MENDELIHT_CV_NO_MERGE");     // (a chain stays with its lane)
    std::vector<DevBuf<double>> yds((size_t)lanes), zds((size_t)lanes);
    PassOrder pass_order;                                      // the lanes' fused passes in single file (common.h)
    // (measured: neutral at configs[3] -- 2.74 s either way with per-fit chains, tools/ab_cv_lanes.sh -- so off unless asked for:
    // MENDELIHT_CV_PASS_ORDER=1, measurement build)
    const bool ordered = lanes > 1 && probe_env("MENDELIHT_CV_PASS_ORDER") != nullptr;
    auto lane = [&](int lane_id) -> int {
        PoolScope from_reserve(h->pool);                        // workspaces and IHTVariables out of the matrix's reserve (no hipMalloc)
        MIH_HIP(hipSetDevice(h->device));
        hipStream_t s = nullptr;
        MIH_TRY(lane_stream_create(h, &s));
        struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{s};
        XtvWork xw; DevBuf<double> R, DF;
        //  the workspace goes back to the matrix's reserve when this scope ends, on error returns too: nothing of this
        // lane may still be queued on the device then, or the other lane would be handed memory that kernels are writing
        struct DrainOnExit { hipStream_t s; ~DrainOnExit() { (void)hipStreamSynchronize(s); } } drain{s};
        MIH_TRY(xtv_work_init(h, xw, cap, tune));
        xw.stream_tag = lane_id + 1;
        hipEvent_t pass_done = nullptr;
        if (ordered) MIH_HIP(hipEventCreateWithFlags(&pass_done, hipEventDisableTiming));
        // (declared behind `drain`: destroyed first -- after this lane's last pass was queued, and an event another stream still waits on may be destroyed)
        struct PassEv { hipEvent_t e; PassOrder *o; ~PassEv() { if (e) { std::lock_guard<std::mutex> g(o->mu); if (o->last == e) o->last = nullptr; (void)hipEventSynchronize(e); (void)hipEventDestroy(e); } } } pass_ev{pass_done, &pass_order};
        if (ordered) { xw.order = &pass_order; xw.pass_done = pass_done; }
        h->prof->count(MIH_CNT_LANES, 1);
        MIH_TRY(R.alloc((size_t)cap * h->n));
        const int init_slots = std::min(init_keys, 8);          // shared initial scores (cv_run_rolling): 8 MB each at p = 1M
        MIH_TRY(DF.alloc((size_t)(cap + init_slots) * h->p));
        DevBuf<double> &yd = yds[(size_t)lane_id], &zd = zds[(size_t)lane_id];   // y and z go up once per lane, not once per fit; they
        MIH_TRY(yd.alloc((size_t)h->n)); MIH_TRY(zd.alloc((size_t)h->n * (size_t)q));  // outlive the lane (its fits may be handed over)
        MIH_HIP(hipMemcpyAsync(yd.p, y_host, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, s));
        MIH_HIP(hipMemcpyAsync(zd.p, z_host, sizeof(double) * (size_t)h->n * (size_t)q, hipMemcpyHostToDevice, s));
        MIH_HIP(hipStreamSynchronize(s));                         // the fits read y and z from their own streams
        IbShared ib_cache;
        CvShared shared; shared.y = yd.p; shared.z = zd.p; shared.ib = &ib_cache;
        if (probe_env("MENDELIHT_CV_NO_COOP") == nullptr)             // A/B (measurement build): round 2's walk, one fit after the other on the lane's stream
            for (int i = 0; i < kWorkerStreamsPerLane; ++i) {
                hipStream_t ws = worker_stream(h, (lane_id % 2) * kWorkerStreamsPerLane + i);
                if (ws) shared.streams.push_back(ws);
            }
        const int rc = cv_run_rolling(h, pr, total, queue, cap, make, xw, R, DF, s, mses_raw, merge_tail ? &handover : nullptr, lane_id, init_slots, shared, lanes == 2 ? inflight : nullptr, chains);
        if (rc == MIH_OK) xtv_count_peels(h, xw, s);
        return rc;
    };
    if (lanes == 1) return lane(0);
    std::vector<int> rcs((size_t)lanes, MIH_OK);
    std::vector<std::string> msgs((size_t)lanes);
    std::vector<std::thread> th;
    for (int g = 0; g < lanes; ++g)
        th.emplace_back([&, g]() {
            rcs[g] = lane(g);
            if (rcs[g]) { char buf[512]; (void)mih_last_error(buf, sizeof(buf)); msgs[g] = buf; queue.halt(); }   // the error text is thread-local
            if (rcs[g] && g == 0) { std::lock_guard<std::mutex> lk(handover.mu); handover.accepting = false; }
        });
    for (auto &t : th) t.join();
    for (int g = 0; g < lanes; ++g)
        if (rcs[g]) { set_error("%s", msgs[g].c_str()); return rcs[g]; }
    return MIH_OK;
}

}  // extern "C"

// Which rank evaluates which (fold, k) combination.  The fits of a rank advance in lock-step, so its time is set by how many
// rounds its LONGEST fit needs and by how many fits ride each round; the iteration count of a fit depends mostly on its model
// size k (5 to 17 at BASELINE configs[3], about the same in every fold).  `index mod world` in fold-major order hands a rank the
// same two or three residues of k in every fold (20 = 4 mod 8): some ranks collect the slow model sizes of all folds.  Instead
// the combinations are dealt out round-robin in the order (k descending, fold ascending): every rank gets a stratified sample
// of the model sizes, 12 or 13 fits each at 100 / 8 ("round-robin by expected cost").  Every fit is independent of
// the rank that runs it, so the losses do not depend on the rule.
void mih::cv_assign(const int64_t *path, int64_t npath, int32_t nfolds, int32_t world, std::vector<int32_t> &rank_of)
{
    const int64_t total = (int64_t)nfolds * npath;
    rank_of.assign((size_t)total, 0);
    const char *e = probe_env("MENDELIHT_CV_ASSIGN");               // measurement build: 0 = round 2's fold-major `index mod world`
    if (e && atoi(e) == 0) { for (int64_t i = 0; i < total; ++i) rank_of[(size_t)i] = (int32_t)(i % world); return; }
    std::vector<int64_t> order((size_t)total);
    for (int64_t i = 0; i < total; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        const int64_t ka = path[a % npath], kb = path[b % npath];
        if (ka != kb) return ka > kb;
        return a / npath < b / npath;
    });
    for (int64_t t = 0; t < total; ++t) rank_of[(size_t)order[(size_t)t]] = (int32_t)(t % world);
}

extern "C" {

int mih_cv_assignment(const int64_t *path, int64_t npath, int32_t nfolds, int32_t world, int32_t *rank_of)
{
    if (!path || !rank_of || npath < 1 || nfolds < 1 || world < 1) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    std::vector<int32_t> r;
    cv_assign(path, npath, nfolds, world, r);
    std::copy(r.begin(), r.end(), rank_of);
    return MIH_OK;
}

int mih_cv_iht(const mih_mat *h, const mih_fit_params *prm, const double *y, const double *z,
               int64_t q, const int32_t *folds, int32_t nfolds, const int64_t *path, int64_t npath,
               int32_t rank, int32_t world, double *mses_raw)
{
    PoolScope from_reserve(h ? h->pool : nullptr);      // device buffers out of the matrix's reserve (DevPool, common.h)
    MIH_TRY(check_params(h, prm, q));
    if (prm->comm) { set_error("cross-validation shards over (fold,k) combinations (rank/world), not over columns"); return MIH_BAD_ARG; }
    if (!y || !z || !folds || !path || !mses_raw || nfolds < 1 || npath < 1 || world < 1 || rank < 0 || rank >= world) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    int64_t n = h->n, kmax = 0;
    for (int64_t i = 0; i < npath; ++i) kmax = std::max(kmax, path[i]);
    if (kmax > h->p) { set_error("Sparsity level in `path` cannot be larger than total number of variables"); return MIH_BAD_ARG; }
    for (int64_t i = 0; i < n; ++i) if (folds[i] < 1 || folds[i] > nfolds) { set_error("folds must be in 1..q"); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(h->device));
    for (int64_t i = 0; i < (int64_t)nfolds * npath; ++i) mses_raw[i] = 0.0;
    mih_fit_params pr = *prm;
    pr.k = kmax; pr.progress = nullptr; pr.choose = nullptr;      // (the caller's tie-break callback is for single fits: see the header)
    std::vector<int32_t> rank_of;
    cv_assign(path, npath, nfolds, world, rank_of);

    // training masks, training-row counts and sums of y: once per fold, not once per (fold, k) fit
    std::vector<std::vector<uint8_t>> fold_train((size_t)nfolds, std::vector<uint8_t>((size_t)n));
    std::vector<int64_t> fold_count((size_t)nfolds, 0);
    std::vector<double> fold_ysum((size_t)nfolds, 0.0);
    {
        auto one_fold = [&](int32_t fold) {
            uint8_t *t = fold_train[(size_t)fold - 1].data();
            int64_t cnt = 0; double ys = 0.0;
            for (int64_t i = 0; i < n; ++i) { t[i] = (folds[i] != fold); if (t[i]) { ys += y[i]; ++cnt; } }      // the order of IhtVar::init_pre's loop
            fold_count[(size_t)fold - 1] = cnt; fold_ysum[(size_t)fold - 1] = ys;
        };
        if (n < 100000 || nfolds < 2) for (int32_t fold = 1; fold <= nfolds; ++fold) one_fold(fold);
        else {                                             // a sweep over n per fold: a few host threads (5 ms -> 1 ms at n = 500k, q = 5)
            std::atomic<int32_t> next_fold{1};
            std::vector<std::thread> th;
            for (int t = 0; t < std::min<int32_t>(nfolds, 8); ++t)
                th.emplace_back([&]() { for (int32_t f = next_fold.fetch_add(1); f <= nfolds; f = next_fold.fetch_add(1)) one_fold(f); });
            for (auto &t : th) t.join();
        }
    }
    // this rank's combinations, fold-major (cross_validation.jl:217-223), in batches
    std::vector<std::pair<int32_t, int64_t>> mine;
    const bool chained = prm->est_r != MIH_ESTR_NONE;
    CvChains chains;
    if (!chained) {
        int64_t combo = 0;
        for (int32_t fold = 1; fold <= nfolds; ++fold)
            for (int64_t ik = 0; ik < npath; ++ik, ++combo)
                if (rank_of[(size_t)combo] == rank) mine.emplace_back(fold, ik);
    } else {
        // The NegBin nuisance parameter travels from one fit of a Julia thread to that thread's next fit (CvChains):
        // `Threads.@threads :static for i in eachindex(combinations)` (cross_validation.jl:100) gives thread t of T the block
        // [t*len + min(t, rem), ...) with len, rem = divrem(total, T).  T = mih_fit_params::cv_threads, 0 = 1 = the single-thread order, the reference's default (with T = nfolds and
        // nfolds | total, e.g. always for the full grid, every fold is one chain); T = 1 is the single-thread order.  A chain is
        // evaluated whole by one rank (chain c by rank c mod world): the losses do not depend on `world`.
        const int64_t total = (int64_t)nfolds * npath, T = prm->cv_threads > 0 ? prm->cv_threads : 1;
        const int64_t len = total / T, rem = total % T;
        int64_t c = 0;
        for (int64_t t = 0; t < T; ++t) {
            const int64_t lo = t * len + std::min(t, rem), cnt = len + (t < rem ? 1 : 0);
            if (cnt == 0) continue;
            if (c++ % world != rank) continue;
            chains.fits.emplace_back();
            for (int64_t i = lo; i < lo + cnt; ++i) { chains.fits.back().push_back(mine.size()); mine.emplace_back((int32_t)(i / npath) + 1, i % npath); }
        }
        chains.r.assign(chains.fits.size(), prm->nb_r);
    }
    if (mine.empty()) return MIH_OK;
    auto make = [&](size_t t, CvFit &f, hipStream_t s, const CvShared &sh) -> int {
        int32_t fold = mine[t].first; int64_t ik = mine[t].second;
        if (f.v) MIH_TRY(f.v->set_k(path[ik]));           // a recycled IHTVariable: v.k = sparsity (cross_validation.jl:110)
        else {
            f.v.reset(new IhtVar());
            MIH_TRY(f.v->create(h, &pr, y, z, q, s, sh.y, sh.z, sh.next_stream()));     // sized for max(path), then
            MIH_TRY(f.v->set_k(path[ik]));
        }
        f.train = fold_train[(size_t)fold - 1].data();
        f.v->train_count = fold_count[(size_t)fold - 1]; f.v->train_ysum = fold_ysum[(size_t)fold - 1]; f.v->train_sums_valid = true;
        f.out_index = (int64_t)(fold - 1) * npath + ik;
        f.init_key = chained ? -1 : fold;                  // (the initial residual of a NegBin fit depends on the r it starts from)
        f.v->ib_shared = sh.ib; f.v->ib_key = fold;        // (a recycled IHTVariable may come from the other lane: re-point it)
        return MIH_OK;
    };
    std::vector<int> keys;                                   // fits of one model size take about as long in every fold (CvQueue)
    if (!chained) for (auto &fk : mine) keys.push_back((int)fk.second);
    return cv_run_lanes(h, pr, mine.size(), make, mses_raw, nfolds, y, z, q, chained ? &chains : nullptr, keys);
}

int mih_fit_iht_path(const mih_mat *h, const mih_fit_params *prm, const double *y, const double *z, int64_t q,
                     const int64_t *path, int64_t npath, int32_t rank, int32_t world,
                     double *logl_out, int64_t *iter_out, double *beta_out, double *c_out)
{
    PoolScope from_reserve(h ? h->pool : nullptr);      // device buffers out of the matrix's reserve (DevPool, common.h)
    MIH_TRY(check_params(h, prm, q));
    if (prm->comm) { set_error("model paths shard over the path entries (rank/world), not over columns"); return MIH_BAD_ARG; }
    if (!y || !z || !path || !logl_out || npath < 1 || world < 1 || rank < 0 || rank >= world) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    int64_t kmax = 0;
    for (int64_t i = 0; i < npath; ++i) { if (path[i] < 0) { set_error("negative model size in path"); return MIH_BAD_ARG; } kmax = std::max(kmax, path[i]); }
    MIH_HIP(hipSetDevice(h->device));
    for (int64_t i = 0; i < npath; ++i) {
        logl_out[i] = 0.0;
        if (iter_out) iter_out[i] = 0;
        if (c_out) for (int64_t l = 0; l < q; ++l) c_out[i * q + l] = 0.0;
    }
    mih_fit_params pr = *prm;
    pr.progress = nullptr; pr.choose = nullptr;
    pr.k = kmax;                                     // IHTVariables are sized for the largest model and re-used along the path
    std::vector<int64_t> mine;
    std::vector<int32_t> rank_of;
    cv_assign(path, npath, 1, world, rank_of);      // the rule of mih_cv_assignment with one fold: largest models first, round-robin
    for (int64_t i = 0; i < npath; ++i) if (rank_of[(size_t)i] == rank) mine.push_back(i);
    if (mine.empty()) return MIH_OK;
    auto slots = [&](CvFit &f, int64_t i) {
        f.full_data = true; f.logl_out = logl_out + i; f.iter_out = iter_out ? iter_out + i : nullptr;
        f.beta_out = beta_out ? beta_out + (size_t)i * h->p : nullptr; f.c_out = c_out ? c_out + (size_t)i * q : nullptr;
    };
    double ysum_all = 0.0;
    for (int64_t i = 0; i < h->n; ++i) ysum_all += y[i];                      // the order of IhtVar::init_pre's loop
    auto make = [&](size_t t, CvFit &f, hipStream_t s, const CvShared &sh) -> int {
        slots(f, mine[t]);
        f.init_key = 0;                                    // every fit of the path starts from the same residual (all rows)
        if (!f.v) {                                        // else: recycled from the lane's pool
            f.v.reset(new IhtVar());
            MIH_TRY(f.v->create(h, &pr, y, z, q, s, sh.y, sh.z, sh.next_stream()));     // sized for max(path)
        }
        f.v->ib_shared = sh.ib; f.v->ib_key = 0;
        f.v->nb_r = pr.nb_r;                               // every fit_iht of the path builds its own IHTVariable (cross_validation.jl:254-258): est_r starts from d.r
        f.v->train_count = h->n; f.v->train_ysum = ysum_all; f.v->train_sums_valid = true;      // (all rows: once per path, not once per fit)
        return f.v->set_k(path[mine[t]]);
    };
    return cv_run_lanes(h, pr, mine.size(), make, nullptr, 1, y, z, q);
}

int mih_cv_iht_multi(const mih_mat *const *hs, int32_t nrep, const mih_fit_params *prm, const double *y,
                     const double *z, int64_t q, const int32_t *folds, int32_t nfolds,
                     const int64_t *path, int64_t npath, double *mses_raw)
{
    if (!hs || nrep < 1 || !mses_raw || nfolds < 1 || npath < 1) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    for (int g = 0; g < nrep; ++g) {
        if (!hs[g]) { set_error("replica %d is NULL", g); return MIH_BAD_ARG; }
        if (hs[g]->n != hs[0]->n || hs[g]->p != hs[0]->p) { set_error("replica %d has different dimensions", g); return MIH_BAD_DIM; }
    }
    const size_t cells = (size_t)nfolds * (size_t)npath;
    std::vector<std::vector<double>> part(nrep, std::vector<double>(cells, 0.0));
    std::vector<int> rcs(nrep, MIH_OK);
    std::vector<std::string> msgs(nrep);
    std::vector<std::thread> th;
    for (int g = 0; g < nrep; ++g)
        th.emplace_back([&, g]() {
            rcs[g] = mih_cv_iht(hs[g], prm, y, z, q, folds, nfolds, path, npath, g, nrep, part[g].data());
            if (rcs[g]) { char buf[512]; (void)mih_last_error(buf, sizeof(buf)); msgs[g] = buf; }   // the error text is thread-local
        });
    for (auto &t : th) t.join();
    for (int g = 0; g < nrep; ++g)
        if (rcs[g]) { set_error("replica %d: %s", g, msgs[g].c_str()); return rcs[g]; }
    for (size_t i = 0; i < cells; ++i) { double s = 0.0; for (int g = 0; g < nrep; ++g) s += part[g][i]; mses_raw[i] = s; }
    return MIH_OK;
}

}  // extern "C"
