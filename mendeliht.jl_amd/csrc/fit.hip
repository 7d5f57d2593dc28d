// fit.hip -- fit_iht! (src/fit.jl:145-207) for ONE univariate fit and the step-by-step sessions: mih_fit_iht, mih_session_*,
// mih_cv_meanloss, and the initialize_beta! regressions shared with the multivariate fit.  One IHTVariable, its kernels and its
// steps (host-driven and resident on the device): fit_state.h; the lock-step drivers of cv_iht / iht_run_many_models:
// fit_lockstep.hip.
//
// What lives where: X (2-bit), y, z, cv_wts, xb, zc, mu, r, df and the projection buffer stay in HBM for the whole fit.  The model
// is k-sparse: (index, value) lists -- in device memory while a fit steps resident (resident.inc), on the host otherwise.
#include "fit_state.h"

namespace mih {

int init_beta_regress_device(const mih_mat *h, const double *w_dev, const double *Y_dev, int m, double N,
                             const double *Sy_host, double *beta_dev, double *icpt_sum_host,
                             DevBuf<double> &red, DevBuf<double> &scal, hipStream_t s, const XtvTune &tune)
{
    const int64_t n = h->n, p = h->p;
    XtvWork xw; DevBuf<double> R, S, icpt, sxxd; DevBuf<uint32_t> M; DevBuf<int32_t> cnt;
    MIH_TRY(xtv_work_init(h, xw, 1 + m, tune));
    MIH_TRY(R.alloc((size_t)(1 + m) * n)); MIH_TRY(S.alloc((size_t)(1 + m) * p)); MIH_TRY(icpt.alloc(p));
    hipLaunchKernelGGL(k_ib_rhs, dim3(nblk(n)), dim3(256), 0, s, Y_dev, w_dev, n, m, R.p);
    MIH_TRY(xtv_device(h, xw, R.p, 1 + m, S.p, s));
    if (h->kind == 0) {
        int64_t nwords = h->n_pad / 16;
        MIH_TRY(M.alloc(nwords)); MIH_TRY(cnt.alloc((size_t)2 * p));
        MIH_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int32_t) * 2 * p, s));
        hipLaunchKernelGGL(k_ib_mask, dim3(nblk(nwords)), dim3(256), 0, s, w_dev, n, nwords, M.p);
        dim3 grid((unsigned)((h->nbp + kIbBpPerBlock - 1) / kIbBpPerBlock), (unsigned)h->ncg);
        hipLaunchKernelGGL(k_ib_counts, grid, dim3(256), 0, s, reinterpret_cast<const uint4 *>(h->X), h->nbp, p, M.p, cnt.p);
    } else {
        MIH_TRY(sxxd.alloc(p));
        if (h->Df) hipLaunchKernelGGL(k_ib_dense_sxx<float>, dim3((unsigned)p), dim3(256), 0, s, h->Df, w_dev, n, p, sxxd.p);
        else hipLaunchKernelGGL(k_ib_dense_sxx<double>, dim3((unsigned)p), dim3(256), 0, s, h->D, w_dev, n, p, sxxd.p);
    }
    const int nsb = 64;
    for (int t = 0; t < m; ++t) {
        hipLaunchKernelGGL(k_ib_solve, dim3(nblk(p)), dim3(256), 0, s, S.p, S.p + (size_t)(1 + t) * p, cnt.p, h->miss_ptr, h->miss_row,
                           w_dev, h->mu, h->sinv, h->kind, h->center, h->scale, h->impute, p, N, Sy_host[t], sxxd.p,
                           beta_dev + (size_t)t * p, icpt.p);
        hipLaunchKernelGGL(k_ib_sum, dim3(nsb), dim3(256), 0, s, icpt.p, p, red.p);
        hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, s, red.p, nsb, 1, scal.p);
        MIH_HIP(hipMemcpyAsync(&icpt_sum_host[t], scal.p, sizeof(double), hipMemcpyDeviceToHost, s));
        MIH_HIP(hipStreamSynchronize(s));
    }
    return MIH_OK;
}

static double sample_var(const double *a, int64_t n)
{
    double m = 0.0; for (int64_t i = 0; i < n; ++i) m += a[i]; m /= (double)n;
    double s = 0.0; for (int64_t i = 0; i < n; ++i) s += (a[i] - m) * (a[i] - m);
    return s / (double)(n - 1);
}

}  // namespace mih

using namespace mih;

extern "C" {

int mih_fit_iht(const mih_mat *h, const mih_fit_params *prm, const double *y, const double *z,
                int64_t q, const uint8_t *train, mih_fit_result *res)
{
    PoolScope from_reserve(h ? h->pool : nullptr);      // device buffers out of the matrix's reserve (DevPool, common.h)
    MIH_TRY(check_params(h, prm, q));
    if (!y || !z || !res) { set_error("null argument"); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(h->device));
    IhtVar v;
    MIH_TRY(v.create(h, prm, y, z, q));
    MIH_TRY(v.init(train));
    auto t0 = std::chrono::steady_clock::now();
    MIH_TRY(v.fit_loop(prm, &res->logl, &res->iter, res->logl_trace, res->tol_trace, res->bt_trace, &res->n_trace));
    res->time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> mu(h->n);
    MIH_HIP(hipMemcpyAsync(mu.data(), v.mu.p, sizeof(double) * h->n, hipMemcpyDeviceToHost, v.s));
    MIH_HIP(hipStreamSynchronize(v.s));
    xtv_count_peels(h, v.xtv, v.s);
    res->pve = sample_var(mu.data(), h->n) / sample_var(y, h->n);   // pve.jl:22,32
    res->nb_r = v.nb_r;
    res->choose_fired = v.choose_fired ? 1 : 0;
    if (res->beta) {
        std::memset(res->beta, 0, sizeof(double) * h->p);
        for (size_t t = 0; t < v.best_b.idx.size(); ++t) res->beta[v.best_b.idx[t]] = v.best_b.val[t];
    }
    if (res->c) for (int l = 0; l < (int)q; ++l) res->c[l] = v.best_c[l];
    if (res->mu) std::memcpy(res->mu, mu.data(), sizeof(double) * h->n);
    return MIH_OK;
}

// ---- cv_iht: rolling lock-step -----------------------------------------------------------
// The (fold, k) fits of cross_validation.jl:100-121 are independent; on one GPU they advance in


struct mih_session_impl {
    IhtVar v;
    mih_fit_params prm;
    double next_logl, best;
    int64_t steps = 0;                  // iht_one_step! calls so far
};

int mih_session_create(const mih_mat *h, const mih_fit_params *prm, const double *y, const double *z,
                       int64_t q, const uint8_t *train, mih_session **out)
{
    PoolScope from_reserve(h ? h->pool : nullptr);      // device buffers out of the matrix's reserve (DevPool, common.h)
    MIH_TRY(check_params(h, prm, q));
    if (!y || !z || !out) { set_error("null argument"); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(h->device));
    mih_session_impl *s = new mih_session_impl();
    s->prm = *prm;
    s->next_logl = s->best = -std::numeric_limits<double>::infinity();
    int rc = s->v.create(h, prm, y, z, q);
    if (!rc) rc = s->v.init(train);
    if (rc) { delete s; return rc; }
    *out = reinterpret_cast<mih_session *>(s);
    return MIH_OK;
}

// `nsteps` iht_one_step! calls.  A fit that qualifies keeps its iterate on the device between calls (IhtVar::res_*): the steps of one
// call are queued back to back, up to two ahead of the records the host reads.  No stopping rule here: the caller decides.
static int session_steps(mih_session_impl *s, int64_t nsteps, double *logl, int64_t *backtracks, double *tol)
{
    IhtVar &v = s->v;
    int64_t total = 0;
    IhtVar::ResRun rr;
    rr.issued = rr.done = s->steps; rr.limit = s->steps + nsteps; rr.max_step = s->prm.max_step;
    double sc = 0.0;
    for (int64_t t = 0; t < nsteps; ++t) {
        bool stepped = false;
        int nbt = 0;
        if (v.res_ok && !v.res_active) {
            if (v.res_begin(s->next_logl, s->best, s->steps, 0, &s->prm) == MIH_OK) rr.issued = rr.done = s->steps;
            else v.res_ok = false;
        }
        if (v.res_active) {
            ResRecord rec; bool aborted = false;
            MIH_TRY(v.res_next(rr, &rec, &aborted));
            if (aborted) MIH_TRY(v.res_end(&s->next_logl, &s->best, true));
            else {
                if (rec.status == RES_STOP_NAN || rec.status == RES_STOP_INF) {
                    MIH_TRY(v.res_end(nullptr, nullptr));
                    if (rec.status == RES_STOP_NAN) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
                    set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL;
                }
                s->next_logl = rec.logl; nbt = rec.nbt; sc = rec.tol; stepped = true;
            }
        }
        if (!stepped) {
            s->best = v.save_prev(s->next_logl, s->best);
            MIH_TRY(v.one_step(s->next_logl, s->prm.max_step, &nbt, &s->next_logl));
            sc = v.check_convergence();
        }
        ++s->steps;
        total += nbt;
    }
    if (logl) *logl = s->next_logl;
    if (backtracks) *backtracks = total;
    if (tol) *tol = sc;
    if (v.h->prof->on && !v.res_active) xtv_count_peels(v.h, v.xtv, v.s);      // (a resident chain runs ahead of the host: counted when it ends)
    return MIH_OK;
}

int mih_session_step(mih_session *ss, double *logl, int32_t *backtracks, double *tol)
{
    if (!ss) return MIH_BAD_ARG;
    mih_session_impl *s = reinterpret_cast<mih_session_impl *>(ss);
    MIH_HIP(hipSetDevice(s->v.h->device));
    int64_t nbt = 0;
    MIH_TRY(session_steps(s, 1, logl, &nbt, tol));
    if (backtracks) *backtracks = (int32_t)nbt;
    return MIH_OK;
}

int mih_session_run(mih_session *ss, int64_t nsteps, double *logl, int64_t *backtracks, double *tol)
{
    if (!ss) return MIH_BAD_ARG;
    mih_session_impl *s = reinterpret_cast<mih_session_impl *>(ss);
    MIH_HIP(hipSetDevice(s->v.h->device));
    if (backtracks) *backtracks = 0;
    if (nsteps <= 0) return MIH_OK;
    return session_steps(s, nsteps, logl, backtracks, tol);
}

int mih_session_model(mih_session *ss, double *beta, double *c)
{
    if (!ss) return MIH_BAD_ARG;
    mih_session_impl *s = reinterpret_cast<mih_session_impl *>(ss);
    MIH_HIP(hipSetDevice(s->v.h->device));
    if (s->v.res_active) MIH_TRY(s->v.res_end(&s->next_logl, &s->best));       // the iterate comes home; the next step takes it back
    if (beta) {
        std::memset(beta, 0, sizeof(double) * s->v.p);
        for (size_t t = 0; t < s->v.b.idx.size(); ++t) beta[s->v.b.idx[t]] = s->v.b.val[t];
    }
    if (c) for (int l = 0; l < s->v.q; ++l) c[l] = s->v.c[l];
    return MIH_OK;
}

int mih_session_destroy(mih_session *ss)
{
    delete reinterpret_cast<mih_session_impl *>(ss);
    return MIH_OK;
}

int mih_cv_meanloss(const double *mses_raw, const int32_t *folds, int64_t n, int32_t nfolds,
                    int64_t npath, double *mse_out)
{
    if (!mses_raw || !folds || !mse_out) return MIH_BAD_ARG;
    std::vector<int64_t> cnt(nfolds, 0);
    for (int64_t i = 0; i < n; ++i) { if (folds[i] < 1 || folds[i] > nfolds) return MIH_BAD_ARG; cnt[folds[i] - 1]++; }
    for (int64_t i = 0; i < npath; ++i) mse_out[i] = 0.0;
    for (int32_t j = 0; j < nfolds; ++j) {                 // cross_validation.jl:312-317
        double wf = (double)cnt[j] / (double)n;
        for (int64_t i = 0; i < npath; ++i) mse_out[i] += mses_raw[i + (int64_t)j * npath] * wf;
    }
    return MIH_OK;
}

}  // extern "C"
