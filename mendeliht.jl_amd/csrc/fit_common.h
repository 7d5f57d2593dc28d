// fit_common.h -- small device helpers shared by the univariate (fit_state.h, fit.hip, fit_lockstep.hip) and multivariate
// (mv.hip) IHT drivers.  Kernels are `static` so each translation unit gets its own copy.
#pragma once
#include "common.h"
#include <vector>
#include <functional>
#include <memory>
#include <string>
#include <ucontext.h>

namespace mih {

// deterministic block sum (fixed tree) of up to NV values per thread
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *out /* NV values, thread 0 */)
{
    __shared__ double red[NV][256];
    #pragma unroll
    for (int k = 0; k < NV; ++k) red[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            #pragma unroll
            for (int k = 0; k < NV; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        #pragma unroll
        for (int k = 0; k < NV; ++k) out[k] = red[k][0];
    }
}

// xtv_digits = -1: may this residual ride the 43-bit format?  max |r| and sum r^2 per 256-row block, then one workgroup: yes iff
// max |r| <= 128 rms(r) (rms over all n rows; held-out rows count as zeros, which only makes the test stricter) -- the format's
// quantum 2^-43 max|r| then stays below 2^-36 rms(r) per entry, ~1e-11 of a column's X'r.  Fixed order: a fit's answer is its own.
constexpr double kAutoDigitsRatio = 128.0;
static __global__ void __launch_bounds__(256)
k_r_guard(const double *__restrict__ r, int64_t n, double *__restrict__ partial /* [blocks][2] */)
{
    __shared__ double smax[256], ssum[256];
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    const double x = i < n ? r[i] : 0.0;
    smax[threadIdx.x] = fabs(x); ssum[threadIdx.x] = x * x;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + k]); ssum[threadIdx.x] += ssum[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = smax[0]; partial[2 * blockIdx.x + 1] = ssum[0]; }
}
static __global__ void __launch_bounds__(256)
k_r_guard_final(const double *__restrict__ partial, int nblocks, int64_t n, double *__restrict__ out /* [0] = 1.0: the 43-bit format will do */)
{
    __shared__ double smax[256], ssum[256];
    double mx = 0.0, sm = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { mx = fmax(mx, partial[2 * b]); sm += partial[2 * b + 1]; }
    smax[threadIdx.x] = mx; ssum[threadIdx.x] = sm;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + k]); ssum[threadIdx.x] += ssum[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double rms = sqrt(ssum[0] / (double)n);
        out[0] = (smax[0] > 0.0 && smax[0] <= kAutoDigitsRatio * rms && smax[0] < 1.0e300) ? 1.0 : 0.0;
    }
}

// second stage: sum `nblocks` rows of NV partials in fixed order
static __global__ void k_final_sum(const double *__restrict__ partial, int nblocks, int nv, double *__restrict__ out)
{
    __shared__ double red[256];
    for (int k = 0; k < nv; ++k) {
        double a = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 256) a += partial[(int64_t)b * nv + k];
        red[threadIdx.x] = a;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
        if (threadIdx.x == 0) out[k] = red[0];
        __syncthreads();
    }
}

// k_final_sum that also brings results home (publish_block): `words` doubles from pub_src, which may contain `out`
// (so `out` is not restrict-qualified: the publishing loads must see the sums stored above)
static __global__ void k_final_sum_pub(const double *__restrict__ partial, int nblocks, int nv, double *out,
                                       const double *pub_src, double *pub_dst_host, uint64_t words, uint64_t *flag_host, uint64_t seq)
{
    __shared__ double red[256];
    for (int k = 0; k < nv; ++k) {
        double a = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 256) a += partial[(int64_t)b * nv + k];
        red[threadIdx.x] = a;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
        if (threadIdx.x == 0) out[k] = red[0];
        __syncthreads();
    }
    __threadfence();
    __syncthreads();
    publish_block(reinterpret_cast<const uint64_t *>(pub_src), reinterpret_cast<uint64_t *>(pub_dst_host), words, flag_host, seq);
}

static __global__ void k_gather(const double *__restrict__ src, const int64_t *__restrict__ idx, int64_t nnz,
                         double *__restrict__ out)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < nnz) out[t] = src[idx[t]];
}
static __global__ void k_mask_to_wts(const uint8_t *__restrict__ m, int64_t n, int invert, double *__restrict__ w)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) w[i] = ((m[i] != 0) != (invert != 0)) ? 1.0 : 0.0;
}

static inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

struct Sparse {                       // a k-sparse p-vector, sorted by index
    std::vector<int64_t> idx;
    std::vector<double> val;
    void clear() { idx.clear(); val.clear(); }
};

// ---- GLM closed forms (GLM.jl / Distributions.jl) --------------------
__device__ __forceinline__ double d_linkinv(int link, double eta)
{
    switch (link) {
    case MIH_LOGIT: return 1.0 / (1.0 + exp(-eta));
    case MIH_LOG:   return exp(eta);
    case MIH_PROBIT:    return 0.5 * erfc(-eta / 1.4142135623730951);
    case MIH_CLOGLOG:   return -expm1(-exp(eta));
    case MIH_CAUCHIT:   return 0.5 + atan(eta) / 3.141592653589793;
    case MIH_INVERSE:   return 1.0 / eta;
    case MIH_INVSQUARE: return 1.0 / sqrt(eta);
    case MIH_SQRT:      return eta * eta;
    default:        return eta;
    }
}
__device__ __forceinline__ double d_mueta(int link, double eta)
{
    switch (link) {
    case MIH_LOGIT: { double e = exp(-fabs(eta)); double f = 1.0 + e; return e / (f * f); }
    case MIH_LOG:   return exp(eta);
    case MIH_PROBIT:    return exp(-0.5 * eta * eta) / 2.5066282746310002;
    case MIH_CLOGLOG:   return exp(eta) * exp(-exp(eta));
    case MIH_CAUCHIT:   return 1.0 / (3.141592653589793 * (1.0 + eta * eta));
    case MIH_INVERSE:   return -1.0 / (eta * eta);
    case MIH_INVSQUARE: { double m = 1.0 / sqrt(eta); return -m * m * m / 2.0; }
    case MIH_SQRT:      return 2.0 * eta;
    default:        return 1.0;
    }
}
__device__ __forceinline__ double d_glmvar(int dist, double mu, double nb_r)
{
    switch (dist) {
    case MIH_BERNOULLI: return mu * (1.0 - mu);
    case MIH_POISSON:   return mu;
    case MIH_NEGBIN:    return mu * (1.0 + mu / nb_r);
    case MIH_GAMMA:     return mu * mu;
    case MIH_INVGAUSS:  return mu * mu * mu;
    default:            return 1.0;
    }
}
__device__ __forceinline__ double d_xlogy(double x, double y) { return x == 0.0 ? 0.0 : x * log(y); }
__device__ __forceinline__ double d_devresid(int dist, double y, double mu, double nb_r)
{
    switch (dist) {
    case MIH_BERNOULLI: return (y == 1.0) ? -2.0 * log(mu) : -2.0 * log1p(-mu);
    case MIH_POISSON:   return 2.0 * (d_xlogy(y, y / mu) - (y - mu));
    case MIH_NEGBIN: {
        double v = 2.0 * (d_xlogy(y, y / mu) + d_xlogy(y + nb_r, (mu + nb_r) / (y + nb_r)));
        return (mu == 0.0) ? nan("") : v;
    }
    case MIH_GAMMA:    return -2.0 * (log(y / mu) - (y - mu) / mu);
    case MIH_INVGAUSS: { double d = y - mu; return d * d / (y * mu * mu); }
    default: { double d = y - mu; return d * d; }
    }
}
// loglik_obs without the Normal branch (utilities.jl:32-43); Normal is closed-form from the deviance
__device__ __forceinline__ double d_loglik_obs(int dist, double y, double mu, double nb_r)
{
    switch (dist) {
    case MIH_BERNOULLI: return (y == 1.0) ? log(mu) : log(1.0 - mu);
    case MIH_POISSON:   return d_xlogy(y, mu) - mu - lgamma(y + 1.0);
    case MIH_NEGBIN: {
        double pp = nb_r / (mu + nb_r);
        return lgamma(nb_r + y) - lgamma(nb_r) - lgamma(y + 1.0) + nb_r * log(pp) + d_xlogy(y, 1.0 - pp);
    }
    default: return 0.0;
    }
}

// linkfun (the inverse of d_linkinv) and GLM.jl's mustart, used by the debias! refit (csrc/debias.hip)
__device__ __forceinline__ double d_linkfun(int link, double mu)
{
    switch (link) {
    case MIH_LOGIT:     return log(mu / (1.0 - mu));
    case MIH_LOG:       return log(mu);
    case MIH_CLOGLOG:   return log(-log1p(-mu));
    case MIH_CAUCHIT:   return tan(3.141592653589793 * (mu - 0.5));
    case MIH_INVERSE:   return 1.0 / mu;
    case MIH_INVSQUARE: return 1.0 / (mu * mu);
    case MIH_SQRT:      return sqrt(mu);
    default:            return mu;
    }
}
__device__ __forceinline__ double d_mustart(int dist, double y)
{
    switch (dist) {
    case MIH_BERNOULLI: return (y + 0.5) / 2.0;
    case MIH_POISSON:   return y + 0.1;
    case MIH_NEGBIN:    return y + (y == 0.0 ? 1.0 / 6.0 : 0.0);
    case MIH_GAMMA:     return y == 0.0 ? 0.1 : y;
    default:            return y;
    }
}

// One host thread, many fits: the per-fit parts of a lock-step round run as coroutines (ucontext) of the lane's thread.  A fit
// that reaches a readback (spin_wait / stream_sync_coop, common.h) yields, and the thread goes on queueing the next fit's
// kernels -- each fit on a stream of its own, so the chains also overlap on the device.  Round 2 walked the fits one after
// the other on the lane's stream: 13 fits x (26 launches + 3 waits) = 3.3 - 3.5 ms between two fused passes of a GPU's share.
// Tasks run to completion even when one fails (their stacks hold objects with destructors); the first error is returned.
struct LaneSched : CoopSched {
    struct Task { ucontext_t ctx; std::function<int()> fn; int rc = MIH_OK; bool done = false; };
    static constexpr size_t kStack = 1u << 20;
    ucontext_t main_ctx;
    Task *cur = nullptr;
    std::vector<std::unique_ptr<char[]>> stacks;           // reused from round to round
    bool enabled = true;
    void yield() override { Task *t = cur; swapcontext(&t->ctx, &main_ctx); }
    static void entry(unsigned lo, unsigned hi)
    {
        Task *t = reinterpret_cast<Task *>(((uintptr_t)hi << 32) | (uintptr_t)lo);
        t->rc = t->fn();
        t->done = true;                                    // returning switches to uc_link = the scheduler
    }
    int run(std::vector<std::function<int()>> &fns)
    {
        int first = MIH_OK;
        std::string first_msg;
        auto note = [&](int rc) {
            if (rc != MIH_OK && first == MIH_OK) { first = rc; char buf[512]; (void)mih_last_error(buf, sizeof(buf)); first_msg = buf; }
        };
        if (!enabled || fns.size() < 2) {
            for (auto &f : fns) { const int rc = f(); note(rc); if (rc) break; }
        } else {
            std::vector<Task> tasks(fns.size());           // fixed size: the contexts hold pointers into it
            while (stacks.size() < fns.size()) stacks.emplace_back(new char[kStack]);
            for (size_t i = 0; i < fns.size(); ++i) {
                Task &t = tasks[i];
                t.fn = fns[i];
                getcontext(&t.ctx);
                t.ctx.uc_stack.ss_sp = stacks[i].get();
                t.ctx.uc_stack.ss_size = kStack;
                t.ctx.uc_link = &main_ctx;
                const uintptr_t q = reinterpret_cast<uintptr_t>(&t);
                makecontext(&t.ctx, reinterpret_cast<void (*)()>(entry), 2, (unsigned)(q & 0xFFFFFFFFu), (unsigned)(q >> 32));
            }
            CoopSched *prev = current_coop();
            current_coop() = this;
            for (size_t left = tasks.size(); left > 0;) {
                for (auto &t : tasks) {
                    if (t.done) continue;
                    cur = &t;
                    swapcontext(&main_ctx, &t.ctx);
                    if (t.done) { --left; note(t.rc); }
                }
            }
            cur = nullptr;
            current_coop() = prev;
        }
        if (first != MIH_OK) set_error("%s", first_msg.c_str());
        return first;
    }
};


}  // namespace mih
