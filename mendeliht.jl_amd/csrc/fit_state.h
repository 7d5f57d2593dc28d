// fit_state.h -- one IHTVariable on the device (src/data_structures.jl:4-43) and everything a step of it runs: the GLM element-wise
// kernels (score!, update_mu!, loglikelihood, iht_stepsize! weights: src/utilities.jl:9-135, 722-764), iht_one_step! host-driven
// (IhtVar::step_pre / step_post) and resident on the device (resident.inc, IhtVar::res_*), init_iht_indices!, initialize_beta!.
// Included by fit.hip (fit_iht!, sessions) and fit_lockstep.hip (cv_iht, model paths): split out of fit.hip in round 6.  Kernels
// have internal linkage (two translation units include this file).
#pragma once
#include "common.h"
#include "peel.h"
#include "fit_common.h"
#include <map>
#include <functional>
#include <atomic>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <cmath>
#include <limits>
#include <memory>
#include <string>
#include <thread>

namespace mih {

constexpr int kMaxQ = 64;
constexpr int kLaneCuReserve = 0;          // CUs a lock-step lane's fused passes leave to the per-fit kernels (lane_stream_create)
struct QVec { double v[kMaxQ]; };

// zc = Z c  (utilities.jl:113), optional clamp (utilities.jl:114-117)
static __global__ void k_zmul(const double *__restrict__ z, int64_t n, int q, QVec c, int clamp20, double *__restrict__ zc)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    for (int l = 0; l < q; ++l) a += z[(int64_t)l * n + i] * c.v[l];
    if (clamp20) a = a < -20.0 ? -20.0 : (a > 20.0 ? 20.0 : a);
    zc[i] = a;
}

// update_mu! (utilities.jl:74-82) fused with deviance (:52-59) and the loglik terms (:9-20).
// partial[b] = { sum w*devresid, A, sum w, B }: A = sum w*loglik_obs for the families whose loglik_obs does not
// involve phi; Normal / Gamma / InverseGaussian need phi = deviance / n first, so their loglikelihood is
// assembled on the host from sums: Gamma A = sum w (log mu + y/mu), B = sum w log y; InverseGaussian
// B = sum w log(2 pi y^3).
static __global__ void __launch_bounds__(256)
k_mu_loglik(const double *__restrict__ xb, const double *__restrict__ zc, const double *__restrict__ y,
            const double *__restrict__ w, int64_t n, int dist, int link, double nb_r, int with_zc,
            double *__restrict__ mu, double *__restrict__ partial)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (i < n) {
        double eta = with_zc ? xb[i] + zc[i] : xb[i];
        double m = d_linkinv(link, eta);
        mu[i] = m;
        double wt = w[i], yi = y[i];
        v[0] = wt * d_devresid(dist, yi, m, nb_r);
        if (dist == MIH_GAMMA) { v[1] = wt * (log(m) + yi / m); v[3] = wt * log(yi); }
        else if (dist == MIH_INVGAUSS) v[3] = wt * log(6.283185307179586 * yi * yi * yi);
        else if (dist != MIH_NORMAL) v[1] = wt * d_loglik_obs(dist, yi, m, nb_r);
        v[2] = wt;
    }
    block_sum<4>(v, partial ? partial + 4ll * blockIdx.x : nullptr);
}

// score! residual (utilities.jl:128-132): r = mueta(eta)/var(mu) * (y-mu) * cv_wts
static __global__ void __launch_bounds__(256)
k_resid(const double *__restrict__ xb, const double *__restrict__ zc, const double *__restrict__ y,
        const double *__restrict__ mu, const double *__restrict__ w, int64_t n,
        int dist, int link, double nb_r, double *__restrict__ r)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double eta = xb[i] + zc[i];
    double m = mu[i];
    r[i] = d_mueta(link, eta) / d_glmvar(dist, m, nb_r) * (y[i] - m) * w[i];
}

// df2 = Z' r (utilities.jl:134): kZtrBlocks workgroups per covariate (grid.y), fixed-order trees
constexpr int kZtrBlocks = 128;
// (the body: covariate l of one fit; block bx of kZtrBlocks)
__device__ __forceinline__ void b_zt_r(int l, int bx, const double *__restrict__ z, const double *__restrict__ r, int64_t n, double *__restrict__ part,
                                       unsigned *__restrict__ done /* [q], zero */, double *__restrict__ out)
{
    const double *zl = z + (int64_t)l * n;
    double v[1] = {0.0};
    for (int64_t i = bx * 256ll + threadIdx.x; i < n; i += 8 * 256ll * kZtrBlocks) {       // eight rows in flight, the sum in the walk's order
        double a8[8], b8[8];
        #pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t iu = i + u * 256ll * kZtrBlocks; a8[u] = iu < n ? zl[iu] : 0.0; b8[u] = iu < n ? r[iu] : 0.0; }
        #pragma unroll
        for (int u = 0; u < 8; ++u) if (i + u * 256ll * kZtrBlocks < n) v[0] += a8[u] * b8[u];
    }
    block_sum<1>(v, part + (int64_t)l * kZtrBlocks + bx);
    if (threadIdx.x == 0) {           // the block that delivers last adds the partials in block order (k_zt_r_final's sum)
        __threadfence();
        if (atomicAdd(&done[l], 1u) == (unsigned)kZtrBlocks - 1) {
            __threadfence();
            double a = 0.0;
            for (int b = 0; b < kZtrBlocks; ++b)
                a += __hip_atomic_load(&part[(int64_t)l * kZtrBlocks + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[l] = a;
            done[l] = 0;
        }
    }
}
static __global__ void __launch_bounds__(256)
k_zt_r(const double *__restrict__ z, const double *__restrict__ r, int64_t n, double *__restrict__ part,
       unsigned *__restrict__ done /* [q], zero */, double *__restrict__ out)
{
    b_zt_r((int)blockIdx.y, (int)blockIdx.x, z, r, n, part, done, out);
}

// NegBin nuisance-parameter sums over all samples (utilities.jl:158-173 MM, :186-194 Newton).
// which = 0: { sum_i sum_{j<y_i} r/(r+j), sum_i log(r/(r+mu_i)) }; which = 1: { dl/dr, d2l/dr2 }
__device__ __forceinline__ double d_digamma(double x)
{
    double r = 0.0;
    while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
    double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132)))));
}
__device__ __forceinline__ double d_trigamma(double x)
{
    double r = 0.0;
    while (x < 6.0) { r += 1.0 / (x * x); x += 1.0; }
    double f = 1.0 / (x * x);
    return r + 1.0 / x + f / 2 + f / x * (1.0 / 6 - f * (1.0 / 30 - f * (1.0 / 42 - f * (1.0 / 30 - f * (5.0 / 66)))));
}
static __global__ void __launch_bounds__(256)
k_nb_sums(const double *__restrict__ y, const double *__restrict__ mu, int64_t n, double r, int which,
          double *__restrict__ partial)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double v[2] = {0.0, 0.0};
    if (i < n) {
        double yi = y[i], mi = mu[i];
        if (which == 0) {
            double num = 0.0;
            for (long long j = 0; j <= (long long)yi - 1; ++j) num += r / (r + (double)j);
            v[0] = num; v[1] = log(r / (r + mi));
        } else {
            v[0] = -(yi + r) / (mi + r) - log(mi + r) + 1.0 + log(r) + d_digamma(r + yi) - d_digamma(r);
            v[1] = (yi + r) / ((mi + r) * (mi + r)) - 2.0 / (mi + r) + 1.0 / r + d_trigamma(r + yi) - d_trigamma(r);
        }
    }
    block_sum<2>(v, partial + 2ll * blockIdx.x);
}

// ---- initialize_beta! (utilities.jl:776-812): the p univariate regressions y ~ 1 + x_j ------------
// right-hand sides of the fused X'R pass: w (-> sum_train x_j) and w.*y (-> x_j'y over the training rows)
static __global__ void k_ib_rhs(const double *__restrict__ y /* m planes */, const double *__restrict__ w, int64_t n, int m,
                         double *__restrict__ R /* 1 + m planes */)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    R[i] = w[i];
    for (int t = 0; t < m; ++t) R[(int64_t)(1 + t) * n + i] = w[i] * y[(int64_t)t * n + i];
}
// bit 2s of word t is set iff row 16t+s is a training row
static __global__ void k_ib_mask(const double *__restrict__ w, int64_t n, int64_t nwords, uint32_t *__restrict__ M)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= nwords) return;
    uint32_t m = 0;
    for (int s = 0; s < 16; ++s) { int64_t i = t * 16 + s; if (i < n && w[i] != 0.0) m |= 1u << (2 * s); }
    M[t] = m;
}
// dosage-1 / dosage-2 counts per column over the training rows (exact integers), tile-major walk
constexpr int kIbBpPerBlock = 64;
static __global__ void __launch_bounds__(256)
k_ib_counts(const uint4 *__restrict__ X, int64_t nbp, int64_t p, const uint32_t *__restrict__ M, int32_t *__restrict__ cnt)
{
    __shared__ int32_t red[2][32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    const int64_t cg = blockIdx.y;
    int32_t c1 = 0, c2 = 0;
    int64_t bp0 = (int64_t)blockIdx.x * kIbBpPerBlock;
    for (int64_t bp = bp0 + wv; bp < bp0 + kIbBpPerBlock && bp < nbp; bp += 4) {
        uint4 v = X[(cg * nbp + bp) * 64 + lane];
        uint32_t d[4] = {v.x, v.y, v.z, v.w};
        #pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            uint32_t mt = M[bp * 8 + (qd >> 1) * 4 + h * 2 + (qd & 1)];
            c1 += __popc(d[qd] & mt); c2 += __popc((d[qd] >> 1) & mt);
        }
    }
    if (threadIdx.x < 64) red[threadIdx.x / 32][threadIdx.x % 32] = 0;
    __syncthreads();
    atomicAdd(&red[0][m], c1); atomicAdd(&red[1][m], c2);
    __syncthreads();
    if (threadIdx.x < 64) {
        int kk = threadIdx.x / 32, mm = threadIdx.x % 32;
        int64_t j = cg * 32 + mm;
        if (j < p && red[kk][mm]) atomicAdd(&cnt[2 * j + kk], red[kk][mm]);
    }
}
// per column: linreg! (utilities.jl:823-842) on the standardized, imputed column restricted to the
// training rows.  Sx, Sxy come from the X'R pass; Sxx from the integer counts.  A failed Cholesky
// leaves the UNSOLVED right-hand side (sum y, x'y) exactly as the reference's `catch` does.
static __global__ void k_ib_solve(const double *__restrict__ Sxv, const double *__restrict__ Sxyv, const int32_t *__restrict__ cnt,
                           const int64_t *__restrict__ miss_ptr, const int32_t *__restrict__ miss_row,
                           const double *__restrict__ w, const double *__restrict__ mu, const double *__restrict__ sinv,
                           int kind, int center, int scale, int impute, int64_t p, double N, double Sy,
                           const double *__restrict__ dense_sxx, double *__restrict__ beta, double *__restrict__ icpt)
{
    int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= p) return;
    double sx = Sxv[j], sxy = Sxyv[j], sxx;
    if (kind == 0) {
        double m = mu[j], s = scale ? sinv[j] : 1.0, cm = center ? m : 0.0;
        double nm = 0.0;
        for (int64_t t = miss_ptr[j]; t < miss_ptr[j + 1]; ++t) nm += (w[miss_row[t]] != 0.0) ? 1.0 : 0.0;
        double c1 = cnt[2 * j], c2 = cnt[2 * j + 1], c0 = N - nm - c1 - c2;
        double xm = ((impute ? m : 0.0) - cm);
        sxx = s * s * (c0 * cm * cm + c1 * (1.0 - cm) * (1.0 - cm) + c2 * (2.0 - cm) * (2.0 - cm) + nm * xm * xm);
    } else sxx = dense_sxx[j];
    double u11 = sqrt(N), u12 = sx / u11, d = sxx - u12 * u12;
    double b0, b1;
    if (!(N > 0.0) || !(d > 0.0)) { b0 = Sy; b1 = sxy; }
    else {
        double u22 = sqrt(d), w1 = Sy / u11, w2 = (sxy - u12 * w1) / u22;
        b1 = w2 / u22; b0 = (w1 - u12 * b1) / u11;
    }
    beta[j] = b1 < -2.0 ? -2.0 : (b1 > 2.0 ? 2.0 : b1);      // clamp!(v.b, -2, 2)
    icpt[j] = b0;
}
// dense design matrix: sum over training rows of x^2 per column
template <typename T>
__global__ void __launch_bounds__(256)
k_ib_dense_sxx(const T *__restrict__ D, const double *__restrict__ w, int64_t n, int64_t p, double *__restrict__ out)
{
    int64_t j = blockIdx.x;
    double v[1] = {0.0};
    for (int64_t i = threadIdx.x; i < n; i += 256) { double x = (double)D[j * n + i]; v[0] += x * x * w[i]; }
    block_sum<1>(v, out + j);
}
static __global__ void __launch_bounds__(256)
k_ib_sum(const double *__restrict__ x, int64_t p, double *__restrict__ part)
{
    double v[1] = {0.0};
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < p; i += 256ll * gridDim.x) v[0] += x[i];
    block_sum<1>(v, part + blockIdx.x);
}
static __global__ void k_ib_full(const double *__restrict__ beta, const double *__restrict__ weight, int64_t p, double *__restrict__ full)
{
    int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j < p) full[j] = weight ? beta[j] * weight[j] : beta[j];
}

// iht_stepsize! tail (utilities.jl:744-756): xgk = (X_S df_S + Z_idc df2_idc) * sqrt(mueta^2/var) * w;
// partial[b] = sum xgk^2
static __global__ void __launch_bounds__(256)
k_stepsize(const double *__restrict__ xgk, const double *__restrict__ z, const double *__restrict__ xb,
           const double *__restrict__ zc, const double *__restrict__ mu, const double *__restrict__ w,
           int64_t n, int q, QVec df2_idc, int dist, int link, double nb_r, double *__restrict__ partial)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    if (i < n) {
        double a = 0.0;
        for (int l = 0; l < q; ++l) a += z[(int64_t)l * n + i] * df2_idc.v[l];
        double g = xgk[i] + a;
        double me = d_mueta(link, xb[i] + zc[i]);
        double sw = sqrt(me * me / d_glmvar(dist, mu[i], nb_r)) * w[i];
        g *= sw;
        v[0] = g * g;
    }
    block_sum<1>(v, partial + blockIdx.x);
}

// the same with df2 still on the device (the step size of the NEXT step is computed speculatively at the end of a step,
// before df2 has travelled to the host): df2_idc[l] = df2_dev[l] where bit l of idc_mask is set
static __global__ void __launch_bounds__(256)
k_stepsize_dev(const double *__restrict__ xgk, const double *__restrict__ z, const double *__restrict__ xb,
               const double *__restrict__ zc, const double *__restrict__ mu, const double *__restrict__ w,
               int64_t n, int q, const double *__restrict__ df2_dev, unsigned long long idc_mask, int dist, int link, double nb_r,
               double *__restrict__ partial)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    if (i < n) {
        double a = 0.0;
        for (int l = 0; l < q; ++l) a += z[(int64_t)l * n + i] * (((idc_mask >> l) & 1ull) ? df2_dev[l] : 0.0);
        double g = xgk[i] + a;
        double me = d_mueta(link, xb[i] + zc[i]);
        double sw = sqrt(me * me / d_glmvar(dist, mu[i], nb_r)) * w[i];
        g *= sw;
        v[0] = g * g;
    }
    block_sum<1>(v, partial + blockIdx.x);
}

// vectorize!(full_b, b, c, weight, zkeep) after the axpy (utilities.jl:258-263,291-315):
// full[j] = eta*df[j]*w_j here; the k support entries are patched by k_scatter_b.
// The covariate tail full[p .. p+qt) (c + eta df2, or Inf for zkeep slots, utilities.jl:264,313-314) rides along as a
// kernel argument instead of a separate host-to-device copy (qt = 0: no tail).
static __global__ void k_grad_full(const double *__restrict__ df, const double *__restrict__ weight, int64_t p,
                            double eta, double *__restrict__ full, QVec tail, int qt)
{
    int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j < qt) full[p + j] = tail.v[j];
    if (j >= p) return;
    double v = eta * df[j];
    full[j] = weight ? v * weight[j] : v;
}
static __global__ void k_scatter_b(const int64_t *__restrict__ idx, const double *__restrict__ val, int64_t nnz,
                            const double *__restrict__ df, const double *__restrict__ weight, double eta,
                            double *__restrict__ full)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    int64_t j = idx[t];
    double v = fma(eta, df[j], val[t]);          // BLAS.axpy!(eta, df, b) is an fma per element
    full[j] = weight ? v * weight[j] : v;
}
// unvectorize! for the gradient at init (utilities.jl:420): df <- projected full / weight
static __global__ void k_scatter_set(const int64_t *__restrict__ idx, const double *__restrict__ val, int64_t nnz, double *__restrict__ out)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < nnz) out[idx[t]] = val[t];
}

static __global__ void k_set_scalar(double *__restrict__ dst, double v) { *dst = v; }
// sum of squares of a short list in list order (the host loop of iht_stepsize!'s numerator, utilities.jl:754, on the device copy)
static __global__ void k_sumsq_seq(const double *__restrict__ v, int64_t cnt, double *__restrict__ dst)
{
    double a = 0.0;
    for (int64_t t = 0; t < cnt; ++t) a += v[t] * v[t];
    *dst = a;
}
static __global__ void k_copy_scalar(double *__restrict__ dst, const double *__restrict__ src) { *dst = *src; }

static __global__ void k_clamp_pm20(double *__restrict__ x, int64_t n)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    x[i] = a < -20.0 ? -20.0 : (a > 20.0 ? 20.0 : a);
}


static __global__ void k_unvec(const double *__restrict__ full, const double *__restrict__ weight, int64_t p,
                        double *__restrict__ df)
{
    int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= p) return;
    df[j] = weight ? full[j] / weight[j] : full[j];
}
// host copies of d_linkinv / d_mueta for the scalar intercept iteration of init_iht_indices!
static double h_linkinv(int link, double eta)
{
    switch (link) {
    case MIH_LOGIT: return 1.0 / (1.0 + std::exp(-eta));
    case MIH_LOG: return std::exp(eta);
    case MIH_PROBIT: return 0.5 * std::erfc(-eta / 1.4142135623730951);
    case MIH_CLOGLOG: return -std::expm1(-std::exp(eta));
    case MIH_CAUCHIT: return 0.5 + std::atan(eta) / 3.141592653589793;
    case MIH_INVERSE: return 1.0 / eta;
    case MIH_INVSQUARE: return 1.0 / std::sqrt(eta);
    case MIH_SQRT: return eta * eta;
    default: return eta;
    }
}
static double h_mueta(int link, double eta)
{
    switch (link) {
    case MIH_LOGIT: { double e = std::exp(-std::fabs(eta)); double f = 1.0 + e; return e / (f * f); }
    case MIH_LOG: return std::exp(eta);
    case MIH_PROBIT: return std::exp(-0.5 * eta * eta) / 2.5066282746310002;
    case MIH_CLOGLOG: return std::exp(eta) * std::exp(-std::exp(eta));
    case MIH_CAUCHIT: return 1.0 / (3.141592653589793 * (1.0 + eta * eta));
    case MIH_INVERSE: return -1.0 / (eta * eta);
    case MIH_INVSQUARE: { double m = 1.0 / std::sqrt(eta); return -m * m * m / 2.0; }
    case MIH_SQRT: return 2.0 * eta;
    default: return 1.0;
    }
}

#include "resident.inc"        // iht_one_step! resident on the device: the kernels of IhtVar::res_*

// initialize_beta! results shared by the fits of a lock-step lane: the p univariate regressions depend on the training rows only
// (the fold), not on the model size, so the first fit of a fold computes them (two extra passes over X) and the other fits of
// that fold in the lane take them from here.  One lane = one host thread: no locking; a fit that finds an entry still being
// computed by another coroutine of its lane yields until it is ready.
struct IbShared {
    struct Entry { int state = 0; DevBuf<double> beta; std::vector<double> c; };      // state 1: being computed, 2: ready
    std::map<int, std::unique_ptr<Entry>> by_key;
};

// One IHTVariable (src/data_structures.jl:4-43), device-resident.
// initialize_beta! regressions (utilities.jl:776-812, multivariate.jl:519-558) for m responses kept as
// planes of n doubles: beta_dev[t][j] = slope of y_t ~ 1 + x_j over the training rows (clamped to +-2),
// icpt_sum[t] = sum_j intercept.  Two extra passes over X: ONE fused (1+m)-RHS X'R (sum x and x'y_t per
// SNP) and a popcount pass (sum x^2 from exact dosage counts).

struct IhtVar {
    std::shared_ptr<DevPool> reserve;             // very first member: the matrix's reserve outlives this variable's blocks
    Arena arena;                                  // the memory outlives every buffer carved out of it
    const mih_mat *h = nullptr;
    int64_t n = 0, p = 0; int q = 0;
    int64_t k = 0, J = 1; std::vector<int64_t> ks;
    int dist = 0, link = 0, est_r = 0; double nb_r = 1.0;
    std::vector<uint8_t> zkeep; int64_t zkeepn = 0;
    const double *y_host = nullptr, *z_host = nullptr;
    int init_beta = 0, debias = 0;
    int (*choose_cb)(void *, int32_t, const int64_t *, int64_t, int64_t, int64_t *) = nullptr;   // mih_fit_params::choose
    void *choose_user = nullptr;
    XtvTune tune;                 // how this fit's X'r passes run (mih_fit_params::xtv_digits)
    hipStream_t s = nullptr;
    // device
    DevBuf<double> y, z, w, xb, zc, mu, r, xgk, df, full, weight, red, scal, gval, ztr;
    DevBuf<unsigned> ztr_done;
    DevBuf<int64_t> sidx; DevBuf<double> sval;   // staging for support lists
    DevBuf<uint8_t> mask;
    XtvWork xtv; XvWork xv; TopkWork topk;
    PinBuf<double> hpin;                          // pinned landing area of the small readbacks
    int nb = 0;                                   // row blocks
    // host
    Sparse b, b0, best_b, idx;                    // idx.val = df on the support
    std::vector<uint8_t> idc, idc0;
    std::vector<double> c, c0, best_c, df2;
    int64_t ntrain = 0;
    bool choose_fired = false;
    bool has_weight = false;

    // column-sharded fit (mih_comm): this process owns columns [col0, col0 + p) of pg; n-vectors replicated
    const mih_comm *comm = nullptr;
    int64_t col0 = 0, pg = 0;
    int comm_fail(int rc) { set_error("communicator callback failed (%d)", rc); return MIH_BAD_ARG; }
    // (measurement hook on: every exchange is timed -- HIP events around a collective queued on this stream, the host clock
    // around one the host waits for -- and kept per kind: mih_profile_exchange)
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    int allreduce_dev(double *buf, int64_t cnt, int op)
    {
        Profile &pf = *h->prof;
        const int kind = cnt == n + 1 ? 0 : 1;
        ExchRecord rec; rec.kind = kind;
        const bool timed = pf.on && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess;
        if (timed) (void)hipEventRecord(rec.e0, s);
        const int nrc = comm_native_allreduce_on_stream(comm, buf, cnt, op, s, h->device);      // the library's own communicator: queued on this stream
        if (nrc >= 0) {
            if (timed) { (void)hipEventRecord(rec.e1, s); std::lock_guard<std::mutex> g(pf.mu); pf.xopen.push_back(rec); }
            return nrc;
        }
        if (timed) { (void)hipEventDestroy(rec.e0); (void)hipEventDestroy(rec.e1); }
        const double t0 = now_ms();
        MIH_HIP(hipStreamSynchronize(s));
        int rc = comm->allreduce(comm->user, buf, cnt, op, 1);
        pf.exch_host(kind, now_ms() - t0);
        return rc ? comm_fail(rc) : MIH_OK;
    }
    int allreduce_host(double *buf, int64_t cnt, int op)
    {
        const double t0 = now_ms();
        int rc = comm->allreduce(comm->user, buf, cnt, op, 0);
        h->prof->exch_host(3, now_ms() - t0);
        return rc ? comm_fail(rc) : MIH_OK;
    }
    // debias! over the column shards (round 6): `v.idx == v.idx0` holds when it holds on every shard; the whole support is the shards'
    // lists one after the other (column blocks in rank order), so one all-gather of [same?, count] tells every shard whether to
    // refit, how large the panel is and where its own columns sit in it.  The panel is summed over the shards (debias.hip) and every
    // shard runs the same refit on the same numbers: the coefficients of its own columns go into b, all of them into the whole model.
    int debias_sharded()
    {
        const double mine[2] = {b.idx == b0.idx ? 1.0 : 0.0, (double)b.idx.size()};
        std::vector<double> all;
        MIH_TRY(allgather_host(mine, 2, all));
        int64_t total = 0, off = 0;
        bool same = true;
        for (int32_t r = 0; r < comm->world; ++r) {
            same = same && all[(size_t)2 * r] != 0.0;
            if (r < comm->rank) off += (int64_t)all[(size_t)2 * r + 1];
            total += (int64_t)all[(size_t)2 * r + 1];
        }
        if (!same || total == 0) return MIH_OK;
        std::vector<double> beta((size_t)total);
        DebiasShard sh;
        sh.k_total = total; sh.k_off = off;
        sh.reduce = [this](double *buf, int64_t cnt) { return allreduce_dev(buf, cnt, 0); };
        MIH_TRY(debias_glm_device(h, b.idx.data(), (int64_t)b.idx.size(), y.p, dist, link, nb_r, beta.data(), s, &sh));
        for (size_t t = 0; t < b.idx.size(); ++t) b.val[t] = beta[(size_t)off + t];
        if (bg_ok && (int64_t)bg.idx.size() == total) bg.val = beta;       // (bg is sorted by global column: the shards' lists in rank order)
        else bg_ok = false;
        return MIH_OK;
    }
    int allgather_host(const double *send, int64_t cnt, std::vector<double> &recv)
    {
        recv.assign((size_t)cnt * comm->world, 0.0);
        const double t0 = now_ms();
        int rc = comm->allgather(comm->user, send, cnt, recv.data());
        h->prof->exch_host(2, now_ms() - t0);
        return rc ? comm_fail(rc) : MIH_OK;
    }

    bool own_stream = true, batched = false;
    // shared_stream != null: this variable is one of a lock-step batch (mih_cv_iht): it runs on the
    // batch's stream and leaves the X'r pass to the batch driver.
    // y_shared / z_shared: device copies of y and z that outlive this variable (a lock-step lane uploads them once for all its fits)
    // fit_stream: a lock-step fit that queues its small kernels on one of the matrix's worker streams instead of the lane's (the
    // chains of a lane's fits overlap on the device); `ev` orders it against the lane's stream around the fused pass
    hipEvent_t ev = nullptr;
    int create(const mih_mat *hh, const mih_fit_params *prm, const double *yh, const double *zh, int64_t qq,
               hipStream_t shared_stream = nullptr, double *y_shared = nullptr, double *z_shared = nullptr, hipStream_t fit_stream = nullptr)
    {
        h = hh; n = h->n; p = h->p; q = (int)qq; y_host = yh; z_host = zh; init_beta = prm->init_beta; tune = xtv_tune(prm);
        reserve = h->pool_owner;
        comm = prm->comm; pg = p; col0 = 0; debias = prm->debias;
        choose_cb = prm->comm ? nullptr : prm->choose; choose_user = prm->choose_user;
        if (comm) {
            if (!comm->allreduce || !comm->allgather || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world ||
                comm->col_offset < 0 || comm->col_offset + p > comm->p_global) {
                set_error("invalid mih_comm (callbacks, rank/world or column range)"); return MIH_BAD_ARG;
            }
            col0 = comm->col_offset; pg = comm->p_global;
        }
        k = prm->k; J = prm->J; dist = prm->dist; link = prm->link; est_r = prm->est_r; nb_r = prm->nb_r;
        if (prm->ks && prm->nks > 0) { ks.assign(prm->ks, prm->ks + prm->nks); k = 0; }
        zkeep.resize(q); zkeepn = 0;
        for (int l = 0; l < q; ++l) { zkeep[l] = prm->zkeep ? (prm->zkeep[l] != 0) : 1; zkeepn += zkeep[l]; }
        if (shared_stream && fit_stream) {
            s = fit_stream; own_stream = false; batched = true;           // the stream belongs to the matrix (worker_stream)
            MIH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        else if (shared_stream) { s = shared_stream; own_stream = false; batched = true; }
        else MIH_HIP(hipStreamCreate(&s));
        nb = (int)nblk(n);
        int64_t kcap = std::max<int64_t>(std::max<int64_t>(J, 1) * k + q, 64) + 1024;
        for (int64_t v : ks) kcap += v;
        // every buffer below comes out of ONE device block and ONE pinned block (Arena, common.h)
        {
            size_t dev = sizeof(double) * ((size_t)n * (7 + q) + 2 * (size_t)p + q + 8) + (size_t)n
                         + sizeof(double) * ((size_t)nb * 4 + kMaxQ + 8 + (size_t)kMaxQ * kZtrBlocks) + sizeof(unsigned) * kMaxQ
                         + 3 * sizeof(double) * (size_t)kcap + xv_work_bytes(h, kcap, kcap - 1024)
                         + sizeof(uint32_t) * 2048 + 64 + 16 * ((size_t)kcap + 1025) + 16 * (4096 + 1) + 256
                         + (prm->weight ? sizeof(double) * (size_t)p : 0) + (prm->group ? sizeof(int64_t) * ((size_t)p + ks.size() + 1) : 0)
                         + 48 * 256
                         + sizeof(ResCtl) + (size_t)kcap * (3 * 16 + 4 * 4) + 4096 * sizeof(uint32_t) + 8 * 256       // resident steps
                         + sizeof(uint64_t) * (size_t)kResCollectBlocks * (1 + 2 * kResCollectSlots);
            size_t pin = sizeof(uint64_t) * (HostStage::kSlots * (2 * (size_t)kcap + 8) + ((size_t)kcap + kMaxQ + 16) + 2 + 2 * ((size_t)kcap + 64) + 16) + 8 * 256
                         + sizeof(ResCtl) + kResRing * sizeof(ResRecord) + (size_t)kcap * (3 * 16 + 4 * 4) + 8 * 256;
            MIH_TRY(arena.reserve(dev, pin));
        }
        ArenaScope in_arena(&arena);
        if (y_shared && z_shared) { y.attach(y_shared, n); z.attach(z_shared, (size_t)n * q); }
        else { MIH_TRY(y.alloc(n)); MIH_TRY(z.alloc((size_t)n * q)); }
        MIH_TRY(w.alloc(n)); MIH_TRY(xb.alloc(n));
        MIH_TRY(zc.alloc(n)); MIH_TRY(mu.alloc(n)); MIH_TRY(r.alloc(n)); MIH_TRY(xgk.alloc((size_t)n + 8));     // (+ the scalar that rides the all-reduce of a sharded fit)
        MIH_TRY(df.alloc(p)); MIH_TRY(full.alloc((size_t)p + q)); MIH_TRY(mask.alloc(n));
        MIH_TRY(red.alloc((size_t)nb * 4)); MIH_TRY(scal.alloc(kMaxQ + 8)); MIH_TRY(ztr.alloc((size_t)kMaxQ * kZtrBlocks));
        MIH_TRY(ztr_done.alloc(kMaxQ)); MIH_HIP(hipMemsetAsync(ztr_done.p, 0, sizeof(unsigned) * kMaxQ, s));      // k_zt_r leaves the counters at zero
        MIH_TRY(sidx.alloc(kcap)); MIH_TRY(sval.alloc(kcap)); MIH_TRY(gval.alloc(kcap));
        MIH_TRY(stage.init(2 * (size_t)kcap + 8));
        MIH_TRY(hpin.alloc((size_t)kcap + kMaxQ + 16, true));
        MIH_TRY(flag.word.alloc(8, true)); flag.word.p[0] = 0; flag.seq = 0;
        if (!batched) { ArenaScope own_buffers(nullptr); MIH_TRY(xtv_work_init(h, xtv, 1, tune, false)); }     // a few large buffers: their own allocations
        MIH_TRY(xv_work_init(h, xv, kcap, kcap - 1024));       // the cache is sized for the model, not for the tie slack of the lists
        MIH_TRY(topk_work_init(topk, kcap));
        if (!(y_shared && z_shared)) {
            MIH_HIP(hipMemcpyAsync(y.p, yh, sizeof(double) * n, hipMemcpyHostToDevice, s));
            MIH_HIP(hipMemcpyAsync(z.p, zh, sizeof(double) * (size_t)n * q, hipMemcpyHostToDevice, s));
        }
        if (prm->weight) {
            has_weight = true;
            MIH_TRY(weight.alloc(p));
            MIH_HIP(hipMemcpyAsync(weight.p, prm->weight, sizeof(double) * p, hipMemcpyHostToDevice, s));
        }
        if (prm->group) {
            has_group = true;
            G = 0;
            for (int64_t j = 0; j < p; ++j) {
                if (prm->group[j] < 1) { set_error("group labels must be 1..G"); return MIH_BAD_ARG; }
                G = std::max(G, prm->group[j]);
            }
            if (comm) {           // a column shard (round 6): the labels of the LOCAL columns; G is the largest label anywhere
                double gmax = (double)G;
                MIH_TRY(allreduce_host(&gmax, 1, 1));
                G = (int64_t)gmax;
                group_host.assign(prm->group, prm->group + p);
            }
            if (!ks.empty() && (int64_t)ks.size() < G) { set_error("k (vector) must have one entry per group"); return MIH_BAD_DIM; }
            MIH_TRY(group_dev.alloc(p));
            MIH_HIP(hipMemcpyAsync(group_dev.p, prm->group, sizeof(int64_t) * p, hipMemcpyHostToDevice, s));
            std::vector<int64_t> kk = ks.empty() ? std::vector<int64_t>{k} : ks;
            MIH_TRY(kgrp_dev.alloc(kk.size()));
            MIH_HIP(hipMemcpy(kgrp_dev.p, kk.data(), sizeof(int64_t) * kk.size(), hipMemcpyHostToDevice));
        } else if (!ks.empty()) {
            set_error("Doubly sparse projection specified (since k is a vector) but there are no group information.");
            return MIH_BAD_ARG;
        }
        c.assign(q, 0.0); c0 = c; best_c = c; df2 = c; idc.assign(q, 0); idc0 = idc;
        MIH_TRY(res_setup(prm, kcap));
        return MIH_OK;
    }
    // the last readback may have been a polled one (SpinFlag): the publishing kernel can still be retiring.  Drain the stream before
    // its buffers go (releasing the arena under a stream that was destroyed with work in flight leaked the block).
    ~IhtVar()
    {
        if (s) (void)hipStreamSynchronize(s);
        if (s && own_stream) (void)hipStreamDestroy(s);
        if (ev) (void)hipEventDestroy(ev);
    }

    // v.k = sparsity (cross_validation.jl:110): with groups and a scalar k the projection reads k from the device
    int set_k(int64_t knew)
    {
        k = knew;
        if (has_group && ks.empty())
            MIH_HIP(hipMemcpyAsync(kgrp_dev.p, &k, sizeof(int64_t), hipMemcpyHostToDevice, s));     // k outlives the copy (member)
        return MIH_OK;
    }

    int ensure_stage(int64_t nnz)
    {
        if ((size_t)nnz <= sidx.n) return MIH_OK;
        MIH_HIP(hipStreamSynchronize(s));
        MIH_TRY(sidx.alloc((size_t)nnz * 2)); MIH_TRY(sval.alloc((size_t)nnz * 2)); MIH_TRY(gval.alloc((size_t)nnz * 2));
        MIH_TRY(stage.init((size_t)nnz * 4));
        stage_forget();
        return MIH_OK;
    }
    // What the staging buffers sidx / sval hold on the device.  A step re-sends the same lists several times (the support of
    // update_xb! again in the gather of the next score and, unless the step backtracked, in the next gradient step): a list
    // that is already there is not sent again, a new one goes through the pinned ring (HostStage) and ONE small kernel.
    HostStage stage;
    std::vector<int64_t> dev_idx; std::vector<double> dev_val; bool dev_idx_ok = false, dev_val_ok = false;
    void stage_forget() { dev_idx_ok = dev_val_ok = false; }
    int upload(const std::vector<int64_t> &ix, const std::vector<double> &vl)
    {
        MIH_TRY(ensure_stage((int64_t)ix.size()));
        if (ix.empty()) return MIH_OK;
        if (dev_idx_ok && dev_val_ok && dev_idx == ix && dev_val.size() == vl.size() &&
            std::memcmp(dev_val.data(), vl.data(), sizeof(double) * vl.size()) == 0) return MIH_OK;       // bits, not values: -0.0 != 0.0 here
        const uint64_t *pin = nullptr;
        MIH_TRY(stage.put(s, ix.data(), sizeof(int64_t) * ix.size(), vl.data(), sizeof(double) * vl.size(), &pin));
        if (pin) stage_to_device(s, pin, reinterpret_cast<uint64_t *>(sidx.p), ix.size(), reinterpret_cast<uint64_t *>(sval.p), vl.size());
        else {
            MIH_HIP(hipMemcpyAsync(sidx.p, ix.data(), sizeof(int64_t) * ix.size(), hipMemcpyHostToDevice, s));
            MIH_HIP(hipMemcpyAsync(sval.p, vl.data(), sizeof(double) * vl.size(), hipMemcpyHostToDevice, s));
        }
        dev_idx = ix; dev_val = vl; dev_idx_ok = dev_val_ok = true;
        return MIH_OK;
    }
    int upload_idx(const std::vector<int64_t> &ix)       // sidx only; sval keeps its content only if the list is unchanged
    {
        MIH_TRY(ensure_stage((int64_t)ix.size()));
        if (ix.empty()) return MIH_OK;
        if (dev_idx_ok && dev_idx == ix) return MIH_OK;
        const uint64_t *pin = nullptr;
        MIH_TRY(stage.put(s, ix.data(), sizeof(int64_t) * ix.size(), nullptr, 0, &pin));
        if (pin) stage_to_device(s, pin, reinterpret_cast<uint64_t *>(sidx.p), ix.size(), nullptr, 0);
        else MIH_HIP(hipMemcpyAsync(sidx.p, ix.data(), sizeof(int64_t) * ix.size(), hipMemcpyHostToDevice, s));
        dev_idx = ix; dev_idx_ok = true; dev_val_ok = false;
        return MIH_OK;
    }
    QVec qvec(const std::vector<double> &v) const { QVec o; for (int l = 0; l < kMaxQ; ++l) o.v[l] = l < q ? v[l] : 0.0; return o; }

    int set_weights(const uint8_t *m, int invert)      // cv_wts from a train mask
    {
        if (!m) {
            std::vector<double> ones(n, invert ? 0.0 : 1.0);
            MIH_HIP(hipMemcpyAsync(w.p, ones.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
            MIH_HIP(hipStreamSynchronize(s));
            return MIH_OK;
        }
        MIH_HIP(hipMemcpyAsync(mask.p, m, n, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_mask_to_wts, dim3(nblk(n)), dim3(256), 0, s, mask.p, n, invert, w.p);
        return MIH_OK;
    }

    // update_xb! (utilities.jl:93-118)
    int update_xb()
    {
        int clamp = (dist != MIH_NORMAL);
        MIH_TRY(upload(b.idx, b.val));
        if (comm) {          // partial X_S b_S of the local support columns, summed over the shards, then the clamp
            MIH_TRY(xv_sparse_device(h, xv, sidx.p, sval.p, (int64_t)b.idx.size(), xb.p, 0, s, b.idx.data(), &stage));
            MIH_TRY(allreduce_dev(xb.p, n, 0));
            if (clamp) hipLaunchKernelGGL(k_clamp_pm20, dim3(nblk(n)), dim3(256), 0, s, xb.p, n);
        } else
        MIH_TRY(xv_sparse_device(h, xv, sidx.p, sval.p, (int64_t)b.idx.size(), xb.p, clamp, s, b.idx.data(), &stage));
        hipLaunchKernelGGL(k_zmul, dim3(nblk(n)), dim3(256), 0, s, z.p, n, q, qvec(c), clamp, zc.p);
        return MIH_OK;
    }
    // update_mu! + loglikelihood; returns logl and the raw deviance
    int mu_loglik(int with_zc, double *logl, double *dev)
    {
        hipLaunchKernelGGL(k_mu_loglik, dim3(nb), dim3(256), 0, s, xb.p, zc.p, y.p, w.p, n, dist, link, nb_r, with_zc, mu.p, red.p);
        MIH_TRY(final_sum_home(4, scal.p, scal.p, 4));
        MIH_HIP(hipGetLastError());                  // a failed launch anywhere in this iteration's chain surfaces here
        const double o[4] = {hpin.p[0], hpin.p[1], hpin.p[2], hpin.p[3]};
        if (dev) *dev = o[0];
        if (logl) {
            if (dist == MIH_NORMAL) {
                double phi = o[0] / (double)n;           // utilities.jl:15: divides by length(y)
                double sd = std::sqrt(phi);
                // sum_i w_i * ( -(z_i^2 + log 2pi)/2 - log sd ),  z_i = (y_i-mu_i)/sd
                *logl = -(o[0] / (sd * sd) + o[2] * 1.8378770664093454835606594728112) / 2.0 - o[2] * std::log(sd);
            } else if (dist == MIH_GAMMA) {          // sum_i w_i logpdf(Gamma(1/phi, mu_i phi), y_i)
                double phi = o[0] / (double)n, a = 1.0 / phi;
                *logl = -o[2] * (std::lgamma(a) + a * std::log(phi)) - a * o[1] + (a - 1.0) * o[3];
            } else if (dist == MIH_INVGAUSS) {       // sum_i w_i logpdf(InverseGaussian(mu_i, 1/phi), y_i)
                double lam = (double)n / o[0];
                *logl = 0.5 * std::log(lam) * o[2] - 0.5 * o[3] - 0.5 * lam * o[0];
            } else *logl = o[1];
        }
        return MIH_OK;
    }
    // score! (utilities.jl:126-135) + df[idx] gather for the next step size
    int score()
    {
        MIH_TRY(resid_only());
        MIH_TRY(xtv_device(h, xtv, r.p, 1, df.p, s));
        return score_post();
    }
    // xtv_digits = -1 in a lock-step driver: does the residual just formed qualify for the 43-bit format?  (GLM links only: the
    // reference's tolerance for them is 1e-4, north_star; Normal / Identity fits keep the 54-bit format.)
    bool auto_digits() const { return batched && tune.digits == -1 && dist != MIH_NORMAL; }
    int residual_rides_43_bits(bool *yes)
    {
        hipLaunchKernelGGL(k_r_guard, dim3(nb), dim3(256), 0, s, r.p, n, red.p);
        hipLaunchKernelGGL(k_r_guard_final, dim3(1), dim3(256), 0, s, red.p, nb, n, scal.p);
        MIH_TRY(readback(scal.p, 1));
        *yes = hpin.p[0] == 1.0;
        return MIH_OK;
    }
    int resid_only()
    {
        hipLaunchKernelGGL(k_resid, dim3(nblk(n)), dim3(256), 0, s, xb.p, zc.p, y.p, mu.p, w.p, n, dist, link, nb_r, r.p);
        return MIH_OK;
    }
    int score_post()       // df is in place (own X'r pass or the batch driver's)
    {
        hipLaunchKernelGGL(k_zt_r, dim3(kZtrBlocks, q), dim3(256), 0, s, z.p, r.p, n, ztr.p, ztr_done.p, scal.p);
        df2_pending = true;                       // lands in pinned memory; copied out at the next synchronisation
        MIH_HIP(hipMemcpyAsync(hpin.p + (hpin.n - kMaxQ), scal.p, sizeof(double) * q, hipMemcpyDeviceToHost, s));
        return MIH_OK;
    }
    // `count` doubles of a device buffer into hpin[0..count) without a stream synchronisation (SpinFlag, common.h)
    SpinFlag flag;
    int readback(const double *src_dev, size_t count)
    {
        MIH_TRY(readback_words(s, flag, reinterpret_cast<const uint64_t *>(src_dev), reinterpret_cast<uint64_t *>(hpin.p), count));
        stage.synced();          // everything queued before has run: the ring's slots are free again
        return MIH_OK;
    }
    // the second stage of a block reduction (nv sums over the nb rows of `red`) and the way home of `count` doubles in ONE kernel
    int final_sum_home(int nv, double *out_dev, const double *src_dev, size_t count)
    {
        const uint64_t seq = spin_begin(flag);
        if (!seq) {
            hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, s, red.p, nb, nv, out_dev);
            return readback(src_dev, count);
        }
        hipLaunchKernelGGL(k_final_sum_pub, dim3(1), dim3(256), 0, s, red.p, nb, nv, out_dev, src_dev, hpin.p, (uint64_t)count, flag.word.p, seq);
        MIH_TRY(spin_wait(s, flag, seq));
        stage.synced();
        return MIH_OK;
    }
    bool df2_pending = false;
    void take_df2() { if (df2_pending) { for (int l = 0; l < q; ++l) df2[l] = hpin.p[hpin.n - kMaxQ + l]; df2_pending = false; } }
    int gather_df_support()
    {
        int64_t nnz = (int64_t)idx.idx.size();
        idx.val.assign(nnz, 0.0);
        if (nnz) {
            MIH_TRY(upload_idx(idx.idx));
            hipLaunchKernelGGL(k_gather, dim3(nblk(nnz)), dim3(256), 0, s, df.p, sidx.p, nnz, gval.p);
            if ((size_t)nnz + kMaxQ > hpin.n) { MIH_HIP(hipStreamSynchronize(s)); take_df2(); MIH_TRY(hpin.alloc((size_t)nnz * 2 + kMaxQ + 16, true)); }
            MIH_HIP(hipMemcpyAsync(hpin.p, gval.p, sizeof(double) * nnz, hipMemcpyDeviceToHost, s));
        }
        MIH_TRY(stream_sync_coop(s));                // (a lane's other fits go on meanwhile)
        for (int64_t t = 0; t < nnz; ++t) idx.val[t] = hpin.p[t];
        take_df2();
        return MIH_OK;
    }

    // the caller draws, as the reference does: `for pos in sample(non_zero_idx, excess, replace=false)` (utilities.jl:453-456)
    int choose_by_caller(Sparse &sp, int64_t excess, bool with_values = true)
    {
        const int64_t nn = (int64_t)sp.idx.size();
        if (excess > nn) { set_error("_choose!: %lld entries to remove out of %lld non-zero SNP effects", (long long)excess, (long long)nn); return MIH_BAD_ARG; }
        std::vector<int64_t> out((size_t)excess, -1);
        if (choose_cb(choose_user, MIH_CHOOSE_SAMPLE, sp.idx.data(), nn, excess, out.data()) != 0) { set_error("the choose callback failed"); return MIH_BAD_ARG; }
        std::vector<char> drop((size_t)nn, 0);
        for (int64_t t = 0; t < excess; ++t) {
            auto it = std::lower_bound(sp.idx.begin(), sp.idx.end(), out[(size_t)t]);
            if (it == sp.idx.end() || *it != out[(size_t)t] || drop[(size_t)(it - sp.idx.begin())]) {
                set_error("the choose callback must return %lld DISTINCT positions out of its list", (long long)excess); return MIH_BAD_ARG;
            }
            drop[(size_t)(it - sp.idx.begin())] = 1;
        }
        Sparse kept;
        for (int64_t i = 0; i < nn; ++i)
            if (!drop[(size_t)i]) { kept.idx.push_back(sp.idx[(size_t)i]); if (with_values) kept.val.push_back(sp.val[(size_t)i]); }
        sp = kept;
        return MIH_OK;
    }

    // _choose! (utilities.jl:444-458): RNG tie-break in the reference; without a callback deterministic here
    // (drop the smallest |b|, ties highest index) and flagged.
    int choose()
    {
        int64_t sparsity = k + zkeepn, groups = (J == 0) ? 1 : J;
        int64_t nsnp = (int64_t)b.idx.size();
        if (comm && bg_ok) nsnp = (int64_t)bg.idx.size();         // every rank holds the whole model of this step (project_full_sharded)
        else if (comm) { double t = (double)nsnp; MIH_TRY(allreduce_host(&t, 1, 0)); nsnp = (int64_t)t; }
        int64_t nz = nsnp - zkeepn;
        for (int l = 0; l < q; ++l) nz += idc[l];
        if (nz <= groups * sparsity) return MIH_OK;
        int64_t excess = nz - groups * sparsity;
        choose_fired = true;
        if (comm && bg_ok) {
            // the deterministic rule on the global list, identically on every rank: the `excess` smallest |b| go (ties: the highest
            // global index first); each shard then keeps what is left of its own columns
            std::vector<size_t> og(bg.idx.size());
            for (size_t i = 0; i < og.size(); ++i) og[i] = i;
            std::sort(og.begin(), og.end(), [&](size_t x, size_t y) {
                const double fx = std::fabs(bg.val[x]), fy = std::fabs(bg.val[y]);
                if (fx != fy) return fx < fy;
                return bg.idx[x] > bg.idx[y];
            });
            std::vector<char> dropg(bg.idx.size(), 0);
            for (int64_t t = 0; t < excess && t < (int64_t)og.size(); ++t) dropg[og[(size_t)t]] = 1;
            Sparse keptg, keptl;
            for (size_t i = 0; i < bg.idx.size(); ++i) {
                if (dropg[i]) continue;
                keptg.idx.push_back(bg.idx[i]); keptg.val.push_back(bg.val[i]);
                const int64_t loc = bg.idx[i] - col0;
                if (loc >= 0 && loc < p) { keptl.idx.push_back(loc); keptl.val.push_back(bg.val[i]); }
            }
            bg = keptg; b = keptl;
            return MIH_OK;
        }
        if (choose_cb) return choose_by_caller(b, excess);
        std::vector<size_t> ord(b.idx.size());
        for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](size_t a, size_t bb) {
            double fa = std::fabs(b.val[a]), fb = std::fabs(b.val[bb]);
            if (fa != fb) return fa < fb;
            return b.idx[a] > b.idx[bb];
        });
        std::vector<char> drop(b.idx.size(), 0);
        if (!comm) {
            for (int64_t t = 0; t < excess && t < (int64_t)ord.size(); ++t) drop[ord[t]] = 1;
        } else {
            // every shard offers its `excess` smallest entries as (|b|, global index); the globally smallest
            // `excess` (ties: highest global index first) are dropped by their owners
            std::vector<double> mine((size_t)excess * 2), all;
            for (int64_t t = 0; t < excess; ++t) {
                bool have = t < (int64_t)ord.size();
                mine[2 * t] = have ? std::fabs(b.val[ord[t]]) : std::numeric_limits<double>::infinity();
                mine[2 * t + 1] = have ? (double)(col0 + b.idx[ord[t]]) : -1.0;
            }
            MIH_TRY(allgather_host(mine.data(), excess * 2, all));
            std::vector<std::pair<double, double>> cand;
            for (size_t t = 0; t + 1 < all.size(); t += 2) if (all[t + 1] >= 0.0) cand.emplace_back(all[t], all[t + 1]);
            std::sort(cand.begin(), cand.end(), [](const std::pair<double, double> &a, const std::pair<double, double> &bb) {
                if (a.first != bb.first) return a.first < bb.first;
                return a.second > bb.second;
            });
            for (int64_t t = 0; t < excess && t < (int64_t)cand.size(); ++t) {
                int64_t g = (int64_t)cand[t].second - col0;
                if (g < 0 || g >= p) continue;
                auto it = std::lower_bound(b.idx.begin(), b.idx.end(), g);
                if (it != b.idx.end() && *it == g) drop[it - b.idx.begin()] = 1;
            }
        }
        Sparse nb2;
        for (size_t i = 0; i < b.idx.size(); ++i) if (!drop[i]) { nb2.idx.push_back(b.idx[i]); nb2.val.push_back(b.val[i]); }
        b = nb2;
        return MIH_OK;
    }

    // project the (p+q) buffer `full` to k+zkeepn and split the survivors into (SNP list, covariate values)
    int project_full(Sparse &snp, std::vector<double> &ctail, std::vector<uint8_t> &ctail_nz, bool zero_in_place = true)
    {
        if (comm) return project_full_sharded(snp, ctail, ctail_nz);
        std::vector<int64_t> si; std::vector<double> sv;
        MIH_TRY(topk_project_device(full.p, p + q, k + zkeepn, topk, s, si, sv, zero_in_place));
        snp.clear();
        ctail_nz.assign(q, 0);
        for (size_t t = 0; t < si.size(); ++t) {
            if (si[t] < p) { snp.idx.push_back(si[t]); snp.val.push_back(sv[t]); }
            else { ctail[si[t] - p] = sv[t]; ctail_nz[si[t] - p] = 1; }
        }
        return MIH_OK;
    }
    // project_k! over the shards: the K-th largest |entry| of the whole vector is the K-th largest of
    // the union of every shard's own top-K (plus the covariate tail, which every rank holds); ties at
    // that value are kept, as in utilities.jl:553-559.  Local survivors of the local projection are a
    // superset of the global survivors because the global threshold is >= every local one.
    // Round 4: a shard sends its candidates as (global index, value) pairs -- 1 + 2K doubles instead of K -- so that every rank ends up
    // with the WHOLE k-sparse model of the step (`bg`): the support count of _choose! and the two maxima of check_convergence are
    // then computed locally and identically on every rank instead of through two more collectives per iteration.  A shard with
    // more than K local survivors (exact ties at its own threshold) can only send K of them: if its smallest sent magnitude
    // still reaches the global threshold the global list may be incomplete, and the step falls back to the collectives (bg_ok).
    // With prior weights the projected values are b * weight (utilities.jl:305-309) and the shards do not hold each other's
    // weights: fallback as well.
    Sparse bg, b0g; bool bg_ok = false, b0g_ok = false;
    int project_full_sharded(Sparse &snp, std::vector<double> &ctail, std::vector<uint8_t> &ctail_nz)
    {
        const int64_t K = k + zkeepn;
        if (K <= 0 || K > pg + q) { set_error("Attempted to project to sparsity level %lld (vector length %lld)", (long long)K, (long long)(pg + q)); return MIH_BAD_ARG; }
        std::vector<double> tail(q);
        MIH_HIP(hipMemcpyAsync(tail.data(), full.p + p, sizeof(double) * q, hipMemcpyDeviceToHost, s));
        std::vector<int64_t> si; std::vector<double> sv;
        const int64_t Kloc = std::min<int64_t>(K, p);
        if (Kloc > 0) MIH_TRY(topk_project_device(full.p, p, Kloc, topk, s, si, sv));
        MIH_HIP(hipStreamSynchronize(s));
        std::vector<size_t> ord(sv.size());
        for (size_t t = 0; t < ord.size(); ++t) ord[t] = t;
        std::sort(ord.begin(), ord.end(), [&](size_t x, size_t y) {
            const double fx = std::fabs(sv[x]), fy = std::fabs(sv[y]);
            if (fx != fy) return fx > fy;
            return si[x] < si[y];
        });
        const int64_t slot = 1 + 2 * K;
        std::vector<double> mine((size_t)slot, -1.0), all;        // index -1 = no entry
        mine[0] = (double)sv.size();
        for (int64_t t = 0; t < K && t < (int64_t)ord.size(); ++t) { mine[1 + 2 * t] = (double)(col0 + si[ord[(size_t)t]]); mine[2 + 2 * t] = sv[ord[(size_t)t]]; }
        MIH_TRY(allgather_host(mine.data(), slot, all));
        std::vector<double> mags;
        for (int32_t r = 0; r < comm->world; ++r)
            for (int64_t t = 0; t < K; ++t) { const double *e = &all[(size_t)r * slot + 1 + 2 * t]; if (e[0] >= 0.0) mags.push_back(std::fabs(e[1])); }
        for (int l = 0; l < q; ++l) mags.push_back(std::fabs(tail[l]));
        if ((int64_t)mags.size() < K) { set_error("projection to %lld entries of a vector with %zu non-empty candidates", (long long)K, mags.size()); return MIH_BAD_ARG; }
        std::nth_element(mags.begin(), mags.begin() + (K - 1), mags.end(), std::greater<double>());
        const double a = mags[K - 1];
        snp.clear();
        ctail_nz.assign(q, 0);
        for (size_t t = 0; t < si.size(); ++t)
            if (std::fabs(sv[t]) >= a) { snp.idx.push_back(si[t]); snp.val.push_back(sv[t]); }
        for (int l = 0; l < q; ++l)
            if (std::fabs(tail[l]) >= a) { ctail[l] = tail[l]; ctail_nz[l] = 1; }
        // the whole model, identical on every rank
        bg.clear(); bg_ok = !has_weight;
        std::vector<std::pair<int64_t, double>> glob;
        for (int32_t r = 0; r < comm->world && bg_ok; ++r) {
            const double *msg = &all[(size_t)r * slot];
            double smallest = std::numeric_limits<double>::infinity();
            for (int64_t t = 0; t < K; ++t) {
                if (msg[1 + 2 * t] < 0.0) continue;
                const double mg = std::fabs(msg[2 + 2 * t]);
                smallest = std::min(smallest, mg);
                if (mg >= a) glob.emplace_back((int64_t)msg[1 + 2 * t], msg[2 + 2 * t]);
            }
            if (msg[0] > (double)K && smallest >= a) bg_ok = false;      // ties cut off at the message size
        }
        if (bg_ok) {
            std::sort(glob.begin(), glob.end());
            for (auto &e : glob) { bg.idx.push_back(e.first); bg.val.push_back(e.second); }
        }
        return MIH_OK;
    }

    // project_group_sparse! (utilities.jl:613-679) over the column shards (round 6).  The reference walks sortperm(|y|, rev = true) -- stable:
    // ties in ascending index -- gives group g its first k_g entries, adds their squares (square rounded, then the sum) to the group's
    // norm in that order, and keeps the J groups of largest norm (ties: the lower label).  A shard's own k_g largest of a group (the
    // device projection with J = G: every group kept) contain whatever of the group's global k_g largest lies in its columns -- a
    // shard's index order is the global one -- so the union of the shards' candidates, walked by the same rule on the host of
    // every rank, gives the reference's survivors.  Two all-gathers (counts, then [global column, value, label] per candidate);
    // y_dev is left projected, `mine` gets the shard's survivors, bg the whole vector's.
    int group_project_sharded(double *y_dev, Sparse &mine)
    {
#pragma clang fp contract(off)
        MIH_TRY(group_project_device(y_dev, group_dev.p, p, G, G, kgrp_dev.p, ks.empty() ? 0 : 1, s));
        std::vector<int64_t> ci; std::vector<double> cv;
        MIH_TRY(collect_nonzero_device(y_dev, p, topk, s, ci, cv));
        const double cnt = (double)ci.size();
        std::vector<double> counts, all;
        MIH_TRY(allgather_host(&cnt, 1, counts));
        int64_t slot = 0;
        for (double c : counts) slot = std::max(slot, (int64_t)c);
        mine.clear(); bg.clear(); bg_ok = true;
        if (slot == 0) return MIH_OK;                              // (nothing but zeros anywhere: y_dev is what it should be)
        std::vector<double> msg((size_t)3 * slot, -1.0);
        for (size_t t = 0; t < ci.size(); ++t) {
            msg[3 * t] = (double)(col0 + ci[t]); msg[3 * t + 1] = cv[t]; msg[3 * t + 2] = (double)group_host[(size_t)ci[t]];
        }
        MIH_TRY(allgather_host(msg.data(), 3 * slot, all));
        struct Cand { int64_t j; double v; int64_t g; };
        std::vector<Cand> u;
        for (int32_t r = 0; r < comm->world; ++r)                  // (rank order = ascending global column)
            for (int64_t t = 0; t < (int64_t)counts[(size_t)r]; ++t) {
                const double *e = &all[((size_t)r * slot + (size_t)t) * 3];
                u.push_back(Cand{(int64_t)e[0], e[1], (int64_t)e[2]});
            }
        std::stable_sort(u.begin(), u.end(), [](const Cand &a, const Cand &c) { return a.j < c.j; });
        std::stable_sort(u.begin(), u.end(), [](const Cand &a, const Cand &c) { return std::fabs(a.v) > std::fabs(c.v); });
        std::vector<double> norm((size_t)G + 1, 0.0);
        std::vector<int64_t> taken((size_t)G + 1, 0);
        std::vector<char> kept(u.size(), 0);
        for (size_t t = 0; t < u.size(); ++t) {
            const int64_t g = u[t].g, kg = ks.empty() ? k : ks[(size_t)g - 1];
            if (taken[(size_t)g] >= kg) continue;
            const double sq = u[t].v * u[t].v;
            norm[(size_t)g] = norm[(size_t)g] + sq;
            ++taken[(size_t)g]; kept[t] = 1;
        }
        std::vector<int64_t> order((size_t)G);
        for (int64_t g = 0; g < G; ++g) order[(size_t)g] = g + 1;
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t c) { return norm[(size_t)a] > norm[(size_t)c]; });
        std::vector<char> in((size_t)G + 1, 0);
        for (int64_t t = 0; t < J && t < G; ++t) in[(size_t)order[(size_t)t]] = 1;
        std::vector<std::pair<int64_t, double>> sv;
        for (size_t t = 0; t < u.size(); ++t) if (kept[t] && in[(size_t)u[t].g]) sv.emplace_back(u[t].j, u[t].v);
        std::sort(sv.begin(), sv.end());
        for (auto &e : sv) {
            bg.idx.push_back(e.first); bg.val.push_back(e.second);
            if (e.first >= col0 && e.first < col0 + p) { mine.idx.push_back(e.first - col0); mine.val.push_back(e.second); }
        }
        MIH_HIP(hipMemsetAsync(y_dev, 0, sizeof(double) * p, s));
        MIH_TRY(upload(mine.idx, mine.val));
        if (!mine.idx.empty())
            hipLaunchKernelGGL(k_scatter_set, dim3(nblk((int64_t)mine.idx.size())), dim3(256), 0, s, sidx.p, sval.p, (int64_t)mine.idx.size(), y_dev);
        return MIH_OK;
    }

    // _iht_gradstep! (utilities.jl:252-280) from base model (bb, cc) with step eta
    int gradstep(const Sparse &bb, const std::vector<double> &cc, double eta)
    {
        if (has_group) {
            // utilities.jl:266-268: project_group_sparse!(v.b, v.group, J, k): no prior weights, and the
            // covariates are not projected in this branch
            hipLaunchKernelGGL(k_grad_full, dim3(nblk(p)), dim3(256), 0, s, df.p, (const double *)nullptr, p, eta, full.p, QVec{}, 0);
            MIH_TRY(upload(bb.idx, bb.val));
            if (!bb.idx.empty())
                hipLaunchKernelGGL(k_scatter_b, dim3(nblk((int64_t)bb.idx.size())), dim3(256), 0, s, sidx.p, sval.p, (int64_t)bb.idx.size(), df.p, (const double *)nullptr, eta, full.p);
            Sparse snp;
            if (comm) MIH_TRY(group_project_sharded(full.p, snp));
            else {
                MIH_TRY(group_project_device(full.p, group_dev.p, p, G, J, kgrp_dev.p, ks.empty() ? 0 : 1, s));
                MIH_TRY(collect_nonzero_device(full.p, p, topk, s, snp.idx, snp.val));
            }
            b = snp;
            for (int l = 0; l < q; ++l) { c[l] = std::fma(eta, df2[l], cc[l]); idc[l] = (c[l] != 0.0); }
            if (ks.empty()) MIH_TRY(choose());         // typeof(k) == Int && _choose!(v)
            idx.idx = b.idx;
            return MIH_OK;
        }
        const double *wp = has_weight ? weight.p : nullptr;
        std::vector<double> cn(q), tail(q);
        for (int l = 0; l < q; ++l) {
            cn[l] = std::fma(eta, df2[l], cc[l]);
            tail[l] = zkeep[l] ? std::numeric_limits<double>::infinity() : cn[l];
        }
        hipLaunchKernelGGL(k_grad_full, dim3(nblk(p)), dim3(256), 0, s, df.p, wp, p, eta, full.p, qvec(tail), q);
        MIH_TRY(upload(bb.idx, bb.val));
        if (!bb.idx.empty())
            hipLaunchKernelGGL(k_scatter_b, dim3(nblk((int64_t)bb.idx.size())), dim3(256), 0, s, sidx.p, sval.p, (int64_t)bb.idx.size(), df.p, wp, eta, full.p);
        Sparse snp; std::vector<double> ct(q, 0.0); std::vector<uint8_t> cnz;
        MIH_TRY(project_full(snp, ct, cnz, /*zero_in_place=*/false));       // only the survivor lists are used
        if (has_weight) {           // unvectorize!: b = full / weight on the survivors
            std::vector<double> hw(snp.idx.size());
            MIH_TRY(ensure_stage((int64_t)snp.idx.size()));
            if (!snp.idx.empty()) {
                stage_forget();
                MIH_HIP(hipMemcpyAsync(sidx.p, snp.idx.data(), sizeof(int64_t) * snp.idx.size(), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_gather, dim3(nblk((int64_t)snp.idx.size())), dim3(256), 0, s, weight.p, sidx.p, (int64_t)snp.idx.size(), gval.p);
                MIH_HIP(hipMemcpyAsync(hw.data(), gval.p, sizeof(double) * hw.size(), hipMemcpyDeviceToHost, s));
                MIH_HIP(hipStreamSynchronize(s));
            }
            for (size_t t = 0; t < hw.size(); ++t) snp.val[t] /= hw[t];
        }
        b = snp;
        for (int l = 0; l < q; ++l) {
            c[l] = zkeep[l] ? cn[l] : (cnz[l] ? ct[l] : 0.0);
            idc[l] = (c[l] != 0.0);
        }
        MIH_TRY(choose());
        idx.idx = b.idx;          // idx = b .!= 0
        return MIH_OK;
    }
    bool has_group = false;
    int64_t G = 0;
    DevBuf<int64_t> group_dev, kgrp_dev;
    std::vector<int64_t> group_host;             // (column shard) the local columns' labels, for the candidates' messages

    // init_iht_indices! (utilities.jl:366-438), init_beta=false
    int init(const uint8_t *train)
    {
        MIH_TRY(init_pre(train));
        MIH_TRY(xtv_device(h, xtv, r.p, 1, df.p, s));
        return init_post();
    }
    int init_pre(const uint8_t *train)
    {
        train_cur = train; spec_ok = false;
        res_ok = res_eligible; lane_queued = false;
        b.clear(); b0.clear(); best_b.clear(); idx.clear();
        bg.clear(); b0g.clear(); bg_ok = b0g_ok = true;        // (b = 0 on every shard)
        std::fill(c.begin(), c.end(), 0.0); c0 = c; best_c = c; std::fill(df2.begin(), df2.end(), 0.0);
        for (int l = 0; l < q; ++l) { idc[l] = zkeep[l]; idc0[l] = zkeep[l]; }
        choose_fired = false;
        MIH_TRY(set_weights(train, 0));
        ntrain = 0; double ybar = 0.0;
        if (train_sums_valid) { ntrain = train_count; ybar = train_ysum; train_sums_valid = false; }      // the lock-step driver has them per fold
        else for (int64_t i = 0; i < n; ++i) if (!train || train[i]) { ybar += y_host[i]; ntrain++; }
        if (ntrain == 0) { set_error("no training samples"); return MIH_BAD_ARG; }
        ybar /= (double)ntrain;
        for (int it = 0; it < 20; ++it) {          // utilities.jl:400-405
            double g1 = h_linkinv(link, c[0]), g2 = h_mueta(link, c[0]);
            double step = (g1 - ybar) / g2;
            step = step < -1.0 ? -1.0 : (step > 1.0 ? 1.0 : step);
            c[0] -= step;
            if (std::fabs(g1 - ybar) < 1e-10) break;
        }
        MIH_HIP(hipMemsetAsync(xb.p, 0, sizeof(double) * n, s));
        hipLaunchKernelGGL(k_zmul, dim3(nblk(n)), dim3(256), 0, s, z.p, n, q, qvec(c), 0, zc.p);   // no clamp at init (utilities.jl:406)
        hipLaunchKernelGGL(k_mu_loglik, dim3(nb), dim3(256), 0, s, xb.p, zc.p, y.p, w.p, n, dist, link, nb_r, 1, mu.p, red.p);
        return resid_only();
    }
    // initialize_beta!(v, cv_idx) + project_k!(v) (utilities.jl:412-414, 776-812, 561-573).  Two extra
    // passes over X: the fused 2-RHS X'R (sum x, x'y per SNP over the training rows) and a popcount
    // pass (sum x^2 from exact dosage counts).  df stays the dense intercept-only gradient.
    // a lock-step lane's cache of the regressions, keyed by the fit's training rows (fold); null outside the lock-step drivers
    IbShared *ib_shared = nullptr; int ib_key = -1;
    int init_beta_phase(const uint8_t *train)
    {
        if (dist != MIH_NORMAL) { set_error("Intializing beta values only work for Gaussian phenotypes! Sorry!"); return MIH_BAD_ARG; }
        // The regressions depend on the training mask only: cv_iht visits the (fold, k) combinations fold-major,
        // so every k of a fold after the first reuses them (two passes over X saved per fit).
        IbShared::Entry *shared_entry = nullptr;
        if (ib_shared && ib_key >= 0) {
            auto &slot_ = ib_shared->by_key[ib_key];
            if (!slot_) slot_.reset(new IbShared::Entry());
            shared_entry = slot_.get();
            while (shared_entry->state == 1 && coop_can_yield()) current_coop()->yield();     // another fit of this lane is computing it
        }
        DevBuf<double> &betad = (shared_entry && shared_entry->state != 1) ? shared_entry->beta : ib_beta;
        const bool from_shared = &betad != &ib_beta;
        const bool reuse = from_shared ? shared_entry->state == 2
                                       : (ib_valid && ((train == nullptr) == ib_train.empty()) &&
                                          (train == nullptr || std::memcmp(ib_train.data(), train, (size_t)n) == 0));
        std::vector<double> &ibc = from_shared ? shared_entry->c : ib_c;
        if (!reuse) {
            struct Computing {        // a failure must not leave the lane's other fits waiting for this entry
                IbShared::Entry *e; bool ok = false;
                ~Computing() { if (e) e->state = ok ? 2 : 0; }
            } computing{from_shared ? shared_entry : nullptr};
            if (from_shared) shared_entry->state = 1;
            else ib_valid = false;
            if (betad.n < (size_t)p) { ArenaScope own_buffer(nullptr); MIH_TRY(betad.alloc(p)); }
            double Sy = 0.0, N = 0.0;
            std::vector<double> ys;
            for (int64_t i = 0; i < n; ++i) if (!train || train[i]) { Sy += y_host[i]; N += 1.0; ys.push_back(y_host[i]); }
            double c0sum = 0.0;
            MIH_TRY(init_beta_regress_device(h, w.p, y.p, 1, N, &Sy, betad.p, &c0sum, red, scal, s, tune));
            ibc.assign(q, 0.0);
            double cov_c0 = 0.0;
            // non-genetic covariates 2..q on the host (utilities.jl:799-806)
            for (int l = 1; l < q; ++l) {
                double sx = 0, sxx = 0, sxy = 0;
                size_t t = 0;
                for (int64_t i = 0; i < n; ++i) if (!train || train[i]) { double xv = z_host[(size_t)l * n + i]; sx += xv; sxx += xv * xv; sxy += xv * ys[t++]; }
                double u11 = std::sqrt(N), u12 = sx / u11, d = sxx - u12 * u12, b0v, b1v;
                if (!(N > 0.0) || !(d > 0.0)) { b0v = Sy; b1v = sxy; }
                else { double u22 = std::sqrt(d), w1 = Sy / u11, w2 = (sxy - u12 * w1) / u22; b1v = w2 / u22; b0v = (w1 - u12 * b1v) / u11; }
                if (comm) cov_c0 += b0v; else c0sum += b0v;
                ibc[l] = b1v;
            }
            // column shard: the intercepts of the SNP regressions of ALL shards (one scalar exchange), the replicated covariates'
            // once, and the divisor counts every SNP column (utilities.jl:808)
            if (comm) { MIH_TRY(allreduce_host(&c0sum, 1, 0)); c0sum += cov_c0; }
            ibc[0] = c0sum / (double)(pg + q - 1);
            for (int l = 0; l < q; ++l) ibc[l] = ibc[l] < -2.0 ? -2.0 : (ibc[l] > 2.0 ? 2.0 : ibc[l]);
            if (!from_shared) { if (train) ib_train.assign(train, train + n); else ib_train.clear(); ib_valid = true; }
            else MIH_TRY(stream_sync_coop(s));         // the shared regressions are complete before any other stream reads them
            computing.ok = true;
        }
        c = ibc;
        c0 = c;
        // project_k!(v): vectorize (weights, Inf for kept covariates), top-(k + zkeepn), unvectorize
        const double *wp = has_weight ? weight.p : nullptr;
        hipLaunchKernelGGL(k_ib_full, dim3(nblk(p)), dim3(256), 0, s, betad.p, wp, p, full.p);
        std::vector<double> tail(q);
        for (int l = 0; l < q; ++l) tail[l] = zkeep[l] ? std::numeric_limits<double>::infinity() : c[l];
        MIH_HIP(hipMemcpyAsync(full.p + p, tail.data(), sizeof(double) * q, hipMemcpyHostToDevice, s));
        Sparse snp; std::vector<double> ct(q, 0.0); std::vector<uint8_t> cnz;
        MIH_TRY(project_full(snp, ct, cnz));
        if (has_weight && !snp.idx.empty()) {
            std::vector<double> hw(snp.idx.size());
            MIH_TRY(ensure_stage((int64_t)snp.idx.size()));
            stage_forget();
            MIH_HIP(hipMemcpyAsync(sidx.p, snp.idx.data(), sizeof(int64_t) * snp.idx.size(), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_gather, dim3(nblk((int64_t)snp.idx.size())), dim3(256), 0, s, weight.p, sidx.p, (int64_t)snp.idx.size(), gval.p);
            MIH_HIP(hipMemcpyAsync(hw.data(), gval.p, sizeof(double) * hw.size(), hipMemcpyDeviceToHost, s));
            MIH_HIP(hipStreamSynchronize(s));
            for (size_t t = 0; t < hw.size(); ++t) snp.val[t] /= hw[t];
        }
        b = snp; b0 = b;
        if (comm) { b0g = bg; b0g_ok = bg_ok; }          // (project_full_sharded left the whole projected model in bg)
        for (int l = 0; l < q; ++l) { if (!zkeep[l]) c[l] = cnz[l] ? ct[l] : 0.0; idc[l] = (c[l] != 0.0); }
        idx.idx = b.idx;
        MIH_TRY(gather_df_support());
        return MIH_OK;
    }
    const uint8_t *train_cur = nullptr;
    // sum of y over the training rows and their count, in the order of init_pre's own loop, when the caller has them already
    // (cv_iht computes them once per fold instead of once per (fold, k) fit: two sweeps over n on the host per fit otherwise)
    bool train_sums_valid = false; int64_t train_count = 0; double train_ysum = 0.0;
    // initialize_beta! results of the last training mask (reused across the k of one CV fold)
    DevBuf<double> ib_beta; std::vector<double> ib_c; std::vector<uint8_t> ib_train; bool ib_valid = false;
    int init_post()
    {
        MIH_TRY(score_post());
        MIH_TRY(stream_sync_coop(s));
        take_df2();
        if (init_beta) return init_beta_phase(train_cur);
        if (!ks.empty()) {
            // utilities.jl:427-429: project_group_sparse!(v.df, group, J, ks); idx is then taken from
            // v.b (all zero) -> empty initial support; idc = trues
            if (comm) { Sparse kept; MIH_TRY(group_project_sharded(df.p, kept)); bg.clear(); bg_ok = true; }     // (the gradient was projected: the model is still 0 everywhere)
            else MIH_TRY(group_project_device(df.p, group_dev.p, p, G, J, kgrp_dev.p, 1, s));
            idx.clear();
            for (int l = 0; l < q; ++l) idc[l] = 1;
            return MIH_OK;
        }
        // vectorize!(full_b, df, df2) ; project_k! ; unvectorize! -> df is replaced by its own projection
        const double *wp = has_weight ? weight.p : nullptr;
        std::vector<double> tail(q);
        for (int l = 0; l < q; ++l) tail[l] = zkeep[l] ? std::numeric_limits<double>::infinity() : df2[l];
        hipLaunchKernelGGL(k_grad_full, dim3(nblk(p)), dim3(256), 0, s, df.p, wp, p, 1.0, full.p, qvec(tail), q);
        Sparse snp; std::vector<double> ct(q, 0.0); std::vector<uint8_t> cnz;
        MIH_TRY(project_full(snp, ct, cnz));
        if (comm) { bg.clear(); bg_ok = true; }            // (what was projected here is the gradient: the model itself is still 0 on every shard)
        if (comm) {          // the device copy was only projected to the LOCAL threshold: rebuild it from the survivors
            MIH_HIP(hipMemsetAsync(full.p, 0, sizeof(double) * p, s));
            MIH_TRY(upload(snp.idx, snp.val));
            if (!snp.idx.empty())
                hipLaunchKernelGGL(k_scatter_set, dim3(nblk((int64_t)snp.idx.size())), dim3(256), 0, s, sidx.p, sval.p, (int64_t)snp.idx.size(), full.p);
        }
        hipLaunchKernelGGL(k_unvec, dim3(nblk(p)), dim3(256), 0, s, full.p, wp, p, df.p);
        for (int l = 0; l < q; ++l) if (!zkeep[l]) df2[l] = cnz[l] ? ct[l] : 0.0;
        idx.idx = snp.idx;
        for (int l = 0; l < q; ++l) idc[l] = zkeep[l];
        // _choose!(v) at init looks at idx with b == 0: it can only fire on exact ties; flag it
        {
            int64_t nsnp = (int64_t)idx.idx.size();
            if (comm) { double t = (double)nsnp; MIH_TRY(allreduce_host(&t, 1, 0)); nsnp = (int64_t)t; }
            int64_t nz = nsnp - zkeepn;
            for (int l = 0; l < q; ++l) nz += idc[l];
            if (nz > ((J == 0) ? 1 : J) * (k + zkeepn)) {
                choose_fired = true;
                int64_t excess = nz - ((J == 0) ? 1 : J) * (k + zkeepn);
                if (choose_cb) {                   // v.b is all zero here: only v.idx[pos] = false has an effect (utilities.jl:454-456)
                    MIH_TRY(choose_by_caller(idx, excess, false));
                } else if (!comm) {
                    for (int64_t t = 0; t < excess && !idx.idx.empty(); ++t) idx.idx.pop_back();
                } else {         // drop the `excess` highest GLOBAL indices
                    std::vector<double> mine((size_t)excess, -1.0), all;
                    for (int64_t t = 0; t < excess && t < (int64_t)idx.idx.size(); ++t)
                        mine[t] = (double)(col0 + idx.idx[idx.idx.size() - 1 - t]);
                    MIH_TRY(allgather_host(mine.data(), excess, all));
                    std::sort(all.begin(), all.end(), std::greater<double>());
                    const double cut = all[excess - 1];       // indices >= cut go
                    while (!idx.idx.empty() && cut >= 0.0 && (double)(col0 + idx.idx.back()) >= cut) idx.idx.pop_back();
                }
            }
        }
        MIH_TRY(gather_df_support());
        return MIH_OK;
    }

    // iht_stepsize! (utilities.jl:722-764)
    int stepsize(double *eta)
    {
        //  Column-sharded fit: the branch taken here decides whether this rank enters the n+1 all-reduce below, so it must
        // be the same on every rank.  spec_ok is: it is set by step_post_fused (every rank runs it in every step) and cleared by
        // init_pre and save_best_model, the only other places that change idx / b / mu -- all of them in lock-step.  The rank-LOCAL
        // comparison of the supports is therefore not part of the decision under comm (it holds whenever spec_ok does).
        if (spec_ok && (comm || (spec_idx == idx.idx && spec_idc == idc))) {      // computed at the end of the previous step
            spec_ok = false;
            double numer = 0.0;
            if (comm) numer = spec_numer_snp;                  // |df_S|^2 over ALL shards came home with the denominator
            else for (size_t t = 0; t < idx.val.size(); ++t) numer += idx.val[t] * idx.val[t];
            for (int l = 0; l < q; ++l) if (idc[l]) numer += df2[l] * df2[l];
            double e = numer / spec_denom;
            if (std::isinf(e) || std::isnan(e)) e = 1e-8;
            *eta = e;
            return MIH_OK;
        }
        spec_ok = false;
        MIH_TRY(upload(idx.idx, idx.val));
        MIH_TRY(xv_sparse_device(h, xv, sidx.p, sval.p, (int64_t)idx.idx.size(), xgk.p, 0, s, idx.idx.data(), &stage));
        std::vector<double> d2(q);
        double numer = 0.0;
        for (size_t t = 0; t < idx.val.size(); ++t) numer += idx.val[t] * idx.val[t];
        if (comm) {
            // the shards' shares of |df_S|^2 ride the all-reduce of X_S g_S as element n of the vector (one collective instead of two)
            // and come home with the denominator
            hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, s, xgk.p + n, numer);
            MIH_TRY(allreduce_dev(xgk.p, n + 1, 0));
            hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(1), 0, s, scal.p + 1, xgk.p + n);
        }
        for (int l = 0; l < q; ++l) d2[l] = idc[l] ? df2[l] : 0.0;
        hipLaunchKernelGGL(k_stepsize, dim3(nb), dim3(256), 0, s, xgk.p, z.p, xb.p, zc.p, mu.p, w.p, n, q, qvec(d2), dist, link, nb_r, red.p);
        MIH_TRY(final_sum_home(1, scal.p, scal.p, comm ? 2 : 1));
        const double denom = hpin.p[0];
        if (comm) numer = hpin.p[1];
        for (int l = 0; l < q; ++l) if (idc[l]) numer += df2[l] * df2[l];
        double e = numer / denom;
        if (probe_env("MENDELIHT_TRACE_ETA"))            // measurement build: what iht_stepsize! divides (tests/test_gpu_parity.py, the 0/0 cases of the sweeps)
            fprintf(stderr, "stepsize: numer %.17g denom %.17g eta %.17g support %zu df2[0] %.17g\n", numer, denom, e, idx.idx.size(), q ? df2[0] : 0.0);
        if (std::isinf(e) || std::isnan(e)) e = 1e-8;
        *eta = e;
        return MIH_OK;
    }

    double save_prev(double cur, double best)        // utilities.jl:702-712
    {
        b0 = b; c0 = c; idc0 = idc;
        b0g = bg; b0g_ok = bg_ok;
        if (cur > best) { best_b = b; best_c = c; }
        return cur > best ? cur : best;
    }
    int save_best_model()                            // utilities.jl:995-1006
    {
        b = best_b; c = best_c; idx.idx = b.idx; spec_ok = false;
        bg_ok = false;                                         // (the best model is kept per shard only)
        for (int l = 0; l < q; ++l) idc[l] = (c[l] != 0.0);
        MIH_TRY(update_xb());
        MIH_TRY(mu_loglik(0, nullptr, nullptr));     // mu = linkinv(xb): genetic part only
        return MIH_OK;
    }
    double check_convergence()                       // utilities.jl:953-957
    {
        double d = 0.0, nbm = 0.0;
        const bool global = comm && bg_ok && b0g_ok;      // both whole models are here: no exchange
        const Sparse &cb = global ? bg : b, &cb0 = global ? b0g : b0;
        size_t i = 0, j = 0;
        while (i < cb.idx.size() || j < cb0.idx.size()) {
            double vb = 0.0, v0 = 0.0;
            if (j >= cb0.idx.size() || (i < cb.idx.size() && cb.idx[i] < cb0.idx[j])) vb = cb.val[i++];
            else if (i >= cb.idx.size() || cb0.idx[j] < cb.idx[i]) v0 = cb0.val[j++];
            else { vb = cb.val[i++]; v0 = cb0.val[j++]; }
            d = std::max(d, std::fabs(vb - v0)); nbm = std::max(nbm, std::fabs(v0));
        }
        if (comm && !global) {
            double two[2] = {d, nbm};
            if (allreduce_host(two, 2, 1)) return std::numeric_limits<double>::quiet_NaN();
            d = two[0]; nbm = two[1];
        }
        for (int l = 0; l < q; ++l) { d = std::max(d, std::fabs(c[l] - c0[l])); nbm = std::max(nbm, std::fabs(c0[l])); }
        return d / (nbm + 1.0);
    }

    // mle_for_r (utilities.jl:141-247): NegBin nuisance parameter by MM or Newton; every sum runs over
    // ALL samples (the reference does not apply cv_wts here), the line search uses loglikelihood(v).
    int nb_sums(int which, double rr, double *out2)
    {
        hipLaunchKernelGGL(k_nb_sums, dim3(nb), dim3(256), 0, s, y.p, mu.p, n, rr, which, red.p);
        MIH_TRY(final_sum_home(2, scal.p, scal.p, 2));       // (in a lock-step lane the other fits go on while this comes home)
        out2[0] = hpin.p[0]; out2[1] = hpin.p[1];
        return MIH_OK;
    }
    int mle_for_r()
    {
        double o[2];
        if (est_r == MIH_ESTR_MM) {                    // update_r_MM (utilities.jl:158-173)
            MIH_TRY(nb_sums(0, nb_r, o));
            nb_r = -o[0] / o[1];
            return MIH_OK;
        }
        // update_r_newton (utilities.jl:180-247)
        double rr = nb_r, new_r = 1.0, stepsz = 1.0;
        const double saved = nb_r;
        auto ll_at = [&](double x, double *val) { nb_r = x; int rc = mu_loglik(1, val, nullptr); return rc; };
        for (int it = 0; it < 100; ++it) {
            MIH_TRY(nb_sums(1, rr, o));
            double inc = (o[1] < 0.0) ? o[0] / o[1] : o[0];
            new_r = rr - stepsz * inc;
            double old_logl, new_logl;
            MIH_TRY(ll_at(rr, &old_logl));
            for (int j = 0; j < 20; ++j) {
                if (new_r <= 0.0) { stepsz /= 2; new_r = rr - stepsz * inc; }
                else {
                    MIH_TRY(ll_at(new_r, &new_logl));
                    if (old_logl >= new_logl) { stepsz /= 2; new_r = rr - stepsz * inc; }
                    else break;
                }
            }
            if (std::fabs(rr - new_r) <= 1e-6) { nb_r = new_r; return MIH_OK; }
            rr = new_r;
        }
        (void)saved;
        nb_r = rr;
        return MIH_OK;
    }

    // iht_one_step! (fit.jl:213-263)
    int one_step(double old_logl, int nstep, int *bt, double *new_logl)
    {
        MIH_TRY(step_pre(old_logl, nstep, bt, new_logl));
        MIH_TRY(xtv_device(h, xtv, r.p, 1, df.p, s));
        return step_post(*new_logl);
    }
    // everything of iht_one_step! before the X'r pass (ends with the working residual in r)
    int step_pre(double old_logl, int nstep, int *bt, double *new_logl)
    {
        double eta;
        MIH_TRY(stepsize(&eta));
        MIH_TRY(gradstep(b, c, eta));
        MIH_TRY(update_xb());
        double logl;
        MIH_TRY(mu_loglik(1, &logl, nullptr));
        if (est_r != MIH_ESTR_NONE) { MIH_TRY(mle_for_r()); MIH_TRY(mu_loglik(1, &logl, nullptr)); }   // fit.jl:235-240
        int es = 0;
        while (old_logl > logl && es < nstep) {       // _iht_backtrack_ (utilities.jl:484-486)
            eta /= 2;
            MIH_TRY(gradstep(b0, c0, eta));           // backtrack! (utilities.jl:959-973)
            MIH_TRY(update_xb());
            MIH_TRY(mu_loglik(1, &logl, nullptr));
            if (est_r != MIH_ESTR_NONE) { MIH_TRY(mle_for_r()); MIH_TRY(mu_loglik(1, &logl, nullptr)); }
            es++;
        }
        *bt = es; *new_logl = logl;
        return resid_only();
    }
    // The end of a step and the beginning of the next in ONE host synchronisation: Z'r (df2), df on the support and --
    // speculatively, it is discarded if the fit stops here -- the whole iht_stepsize! of the next step (X_S df_S straight
    // from the device copy of df_S, the weighted sum of squares) are queued back to back and come home in one copy:
    // [df_S | df2 | sum xgk^2].  One synchronisation and three small copies less per iteration than doing the step size
    // on its own.
    // Column-sharded fit: the shards' partial X_S df_S and their shares of |df_S|^2 (element n of the vector) are summed by ONE
    // all-reduce in the middle of the chain -- queued on this stream when the communicator is the library's own -- so a sharded step
    // ends with one wait like an unsharded one.
    bool spec_ok = false; double spec_denom = 0.0, spec_numer_snp = 0.0; std::vector<int64_t> spec_idx; std::vector<uint8_t> spec_idc;
    int step_post_fused()
    {
        const int64_t nnz = (int64_t)idx.idx.size();
        MIH_TRY(ensure_stage(nnz + kMaxQ + 3));
        hipLaunchKernelGGL(k_zt_r, dim3(kZtrBlocks, q), dim3(256), 0, s, z.p, r.p, n, ztr.p, ztr_done.p, gval.p + nnz);     // df2 behind df_S
        if (nnz) MIH_TRY(upload_idx(idx.idx));
        MIH_TRY(xv_sparse_device(h, xv, sidx.p, gval.p, nnz, xgk.p, 0, s, idx.idx.data(), &stage, df.p, gval.p));    // df_S gathered on the way
        if (comm) {
            hipLaunchKernelGGL(k_sumsq_seq, dim3(1), dim3(1), 0, s, gval.p, nnz, xgk.p + n);
            MIH_TRY(allreduce_dev(xgk.p, n + 1, 0));
            hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(1), 0, s, gval.p + nnz + q + 1, xgk.p + n);
        }
        unsigned long long mask = 0ull;
        for (int l = 0; l < q; ++l) if (idc[l]) mask |= 1ull << l;
        hipLaunchKernelGGL(k_stepsize_dev, dim3(nb), dim3(256), 0, s, xgk.p, z.p, xb.p, zc.p, mu.p, w.p, n, q, gval.p + nnz, mask,
                           dist, link, nb_r, red.p);
        const size_t home = (size_t)(nnz + q + 1 + (comm ? 1 : 0));
        if (home > hpin.n) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(hpin.alloc((size_t)nnz * 2 + kMaxQ + 16, true)); }
        MIH_TRY(final_sum_home(1, gval.p + nnz + q, gval.p, home));
        idx.val.assign(hpin.p, hpin.p + nnz);
        for (int l = 0; l < q; ++l) df2[l] = hpin.p[nnz + l];
        df2_pending = false;
        spec_denom = hpin.p[nnz + q]; spec_idx = idx.idx; spec_idc = idc; spec_ok = true;
        if (comm) spec_numer_snp = hpin.p[nnz + q + 1];
        return MIH_OK;
    }
    int step_post(double logl)
    {
        MIH_TRY(step_post_fused());
        if (std::isnan(logl)) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
        if (std::isinf(logl)) { set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL; }
        return MIH_OK;
    }


    // ---- iht_one_step! resident on the device (resident.inc) ------------------------------------------------------------------
    // The host-driven members above stay the library's statement of the step (and serve every fit the resident chain does not:
    // column shards, groups, est_r, debias, dense matrices, lock-step lanes); a fit that qualifies runs its steps through
    // res_next().  Between res_begin() and res_end() the iterate lives on the device only.
    bool res_ok = false, res_active = false, res_zero_list = false;
    bool res_eligible = false;           // what res_setup found; res_ok falls to false when a res_begin declines (lists beyond the buffers) and comes back with the next fit on this variable (init_pre)
    int res_epoch = 0; uint64_t res_seq = 0; int64_t res_kcap = 0;
    std::vector<uint64_t> res_out;                       // sequence numbers of the steps in flight, oldest first
    DevBuf<ResCtl> rctl; DevBuf<int64_t> ridx; DevBuf<double> rval; DevBuf<int32_t> rslot; DevBuf<uint32_t> rhist; DevBuf<uint64_t> rsel; DevBuf<double> rwalk; DevBuf<uint32_t> rtick;
    bool res_big = false; DevBuf<uint64_t> rbig;            // a model beyond kResMaxList entries: k_res_select_big and its scratch
    bool res_sharded = false; DevBuf<int64_t> rgidx; DevBuf<double> rgval, rmsg; PinBuf<double> rg_h;        // column shard: whole models, messages
    PinBuf<ResCtl> rctl_h; PinBuf<ResRecord> rrec; PinBuf<int64_t> ridx_h; PinBuf<double> rval_h; PinBuf<int32_t> rslot_h;
    struct ResRun { int64_t limit = 0, issued = 0, done = 0; int max_step = 3; };

    int res_setup(const mih_fit_params *prm, int64_t kcap)
    {
        res_ok = res_eligible = false;
        const int64_t K = k + zkeepn;
        // (round 6) a lock-step lane's fit runs resident too: its chain is queued behind the lane's fused pass (lane_queue_step) and
        // stops at the working residual; the score itself stays the lane's
        if (h->kind != 0 || has_group || !ks.empty() || est_r != MIH_ESTR_NONE || debias || prm->step_mode != 0) return MIH_OK;
        if (probe_env("MENDELIHT_NO_RESIDENT")) return MIH_OK;              // measurement build: A/B against the host-driven step
        // a column shard: only with the library's own communicator (its collectives are queued INSIDE the gated chain; callbacks of
        // the host language need the host), no prior weights (the shards do not hold each other's), and a pool of candidates
        // (world x K and the covariate tail) the second stage of the select can rank
        res_sharded = comm != nullptr;
        if (comm && (!comm_is_native(comm, h->device) || has_weight || (int64_t)comm->world * K + q > kResMaxInBin || K + 64 > kResShardList)) return MIH_OK;
        // (round 6) a model beyond ~2000 effects ranks and orders its survivors in a scratch block of device memory (k_res_select_big)
        // instead of LDS; beyond ~8000, in a column shard or in a lane's batched chain it takes the host-driven step
        res_big = K + 64 > kResMaxList;
        if (res_big && (K + 64 > kResBigList || comm || (batched && probe_env("MENDELIHT_LANE_BATCHED")))) return MIH_OK;
        if (xv.slots <= 0 || xv.slots > (res_big ? 2 * kResBigList + 1024 : 160 * 32) || K < 1 || K > pg + q || h->p >= (1ll << 40)) return MIH_OK;
        res_kcap = std::min<int64_t>(kcap, (int64_t)xv.coefA.n);
        MIH_TRY(rctl.alloc(1)); MIH_TRY(ridx.alloc((size_t)res_kcap * 3)); MIH_TRY(rval.alloc((size_t)res_kcap * 3));
        MIH_TRY(rslot.alloc((size_t)res_kcap * 4)); MIH_TRY(rhist.alloc(4096));
        MIH_TRY(rsel.alloc((size_t)kResCollectBlocks * (1 + 2 * kResCollectSlots)));
        if (res_big) { ArenaScope own_buffer(nullptr); MIH_TRY(rbig.alloc((size_t)kResBigScratchWords)); }
        if (!batched) {             // (a lane's fit has no score of its own: no k_res_stats)
            const size_t slices = (size_t)(q + kResStatCov - 1) / kResStatCov;
            MIH_TRY(rwalk.alloc(slices * kStatBlocksRes * 10 * 256)); MIH_TRY(rtick.alloc(slices * kStatBlocksRes));      // k_res_stats: the walkers' sums, a ticket per walk-block
            MIH_HIP(hipMemsetAsync(rtick.p, 0, sizeof(uint32_t) * slices * kStatBlocksRes, s));
        }
        if (res_sharded) {
            const size_t mlen = 2 + 2 * (size_t)K;
            MIH_TRY(rgidx.alloc((size_t)res_kcap * 3)); MIH_TRY(rgval.alloc((size_t)res_kcap * 3));          // the two whole models + the shard's own survivors
            MIH_TRY(rmsg.alloc(mlen * (size_t)(comm->world + 1)));
            MIH_TRY(rg_h.alloc((size_t)res_kcap * 2, true));
        }
        MIH_TRY(rctl_h.alloc(1, true)); MIH_TRY(rrec.alloc(kResRing, true));
        MIH_TRY(ridx_h.alloc((size_t)res_kcap * 3, true)); MIH_TRY(rval_h.alloc((size_t)res_kcap * 3, true)); MIH_TRY(rslot_h.alloc((size_t)res_kcap * 4, true));
        MIH_HIP(hipMemsetAsync(rhist.p, 0, sizeof(uint32_t) * 4096, s));
        std::memset(rrec.p, 0, sizeof(ResRecord) * kResRing);
        res_ok = res_eligible = true;
        return MIH_OK;
    }
    ResPtrs res_ptrs() const
    {
        ResPtrs P;
        P.ctl = rctl.p;
        for (int i = 0; i < 3; ++i) { P.idx[i] = ridx.p + (size_t)i * res_kcap; P.val[i] = rval.p + (size_t)i * res_kcap; }
        for (int i = 0; i < 2; ++i) { P.slot[i] = rslot.p + (size_t)i * res_kcap; P.fresh[i] = rslot.p + (size_t)(2 + i) * res_kcap; }
        P.gval = gval.p; P.coefA = xv.coefA.p; P.coefB = xv.coefB.p;
        P.hist = rhist.p; P.sel = rsel.p; P.sel_cap = 0; P.kcap = res_kcap; P.rec = rrec.p; P.big = res_big ? rbig.p : nullptr;
        P.gidx[0] = P.gidx[1] = P.lc_idx = nullptr; P.gvals[0] = P.gvals[1] = P.lc_val = P.msg = P.msgs = nullptr; P.world = 1; P.rank = 0; P.col0 = 0;
        if (res_sharded) {
            for (int i = 0; i < 2; ++i) { P.gidx[i] = rgidx.p + (size_t)i * res_kcap; P.gvals[i] = rgval.p + (size_t)i * res_kcap; }
            P.lc_idx = rgidx.p + 2 * (size_t)res_kcap; P.lc_val = rgval.p + 2 * (size_t)res_kcap;
            P.msg = rmsg.p; P.msgs = rmsg.p + (2 + 2 * (size_t)(k + zkeepn));
            P.world = comm->world; P.rank = comm->rank; P.col0 = col0;
        }
        return P;
    }
    ResMat res_mat() const
    {
        ResMat M;
        M.X = h->X; M.nbp = h->nbp; M.ndw = h->n_pad / 16; M.n = n; M.p = p;
        M.cache = xv.cache.p; M.slots = (int32_t)xv.slots;
        M.mu = h->mu; M.sinv = h->sinv; M.center = h->center; M.scale = h->scale;
        M.miss_ptr = h->miss_ptr; M.miss_row = h->miss_row;
        return M;
    }
    // k_res_xgk / k_res_xb: workgroups of 1024 rows (64 dwords of every cached column), 68 KB of dynamic LDS
    unsigned res_wide_blocks() const { return (unsigned)((h->n_pad / 16 + 63) / 64); }
    uint64_t res_zkeep_mask() const { uint64_t m = 0; for (int l = 0; l < q; ++l) if (zkeep[l]) m |= 1ull << l; return m; }
    bool res_fix() const { return h->impute && h->total_missing > 0; }
    // Normal / identity on the plain (unsharded, no imputed entries) path: the attempts do not store xb, zc, mu (k_res_xb's `lean`);
    // res_end forms them again from the model that comes home
    bool res_lean() const { return !res_sharded && !res_fix() && dist == MIH_NORMAL && link == MIH_IDENTITY; }
    void xv_cache_forget()           // the host's map of the column cache no longer describes it (the device kept the books), or vice versa
    {
        xv.slot_of.clear(); std::fill(xv.col_of.begin(), xv.col_of.end(), (int64_t)-1); std::fill(xv.stamp.begin(), xv.stamp.end(), (uint64_t)0); xv.tick = 0;
    }

    // the host-side iterate -> the device.  next_logl / best: the loglikelihoods fit_iht! carries (fit.jl:163-164)
    int res_begin(double next_logl, double best, int64_t iter_done, int arm_stop, const mih_fit_params *prm)
    {
        if (!res_ok) return MIH_BAD_ARG;
        // the list iht_stepsize! and _iht_gradstep! work on: b's support; after init_iht_indices! b is still zero and the list is the
        // support of the projected gradient (utilities.jl:432) -- then those entries ride as explicit zeros of b
        const bool zero_list = b.idx.empty() && !idx.idx.empty();
        if (!zero_list && b.idx != idx.idx) return MIH_BAD_ARG;
        const std::vector<int64_t> &lst = zero_list ? idx.idx : b.idx;
        const int64_t cnt = (int64_t)lst.size(), cb = (int64_t)best_b.idx.size();
        bool fits = !(cnt > res_kcap || cnt > xv.slots || cb > res_kcap);
        if (res_sharded) {            // every shard or none: the chain's collectives must be issued by all (one exchange of a flag)
            fits = fits && bg_ok && (int64_t)bg.idx.size() <= res_kcap;
            double decline = fits ? 0.0 : 1.0;
            MIH_TRY(allreduce_host(&decline, 1, 1));
            fits = decline == 0.0;
        }
        if (!fits) return MIH_BAD_ARG;
        MIH_TRY(stream_sync_coop(s));                        // the pinned staging below may still be read by an earlier upload (a lane's other fits go on meanwhile)
        ResCtl &C = *rctl_h.p;
        std::memset(&C, 0, sizeof(C));
        C.live_epoch = res_gate_step(res_epoch); C.cur = 0; C.es = 0; C.iter = (int32_t)iter_done;
        C.arm_stop = arm_stop; C.min_iter = prm->min_iter; C.max_step = prm->max_step; C.tol_stop = prm->tol;
        C.logl_cur = next_logl; C.best_logl = best;
        for (int l = 0; l < q; ++l) C.df2[l] = df2[l];
        C.m[0].cnt = cnt; C.best.cnt = cb;
        for (int l = 0; l < q; ++l) { C.m[0].c[l] = c[l]; if (idc[l]) C.m[0].idc |= 1ull << l; C.best.c[l] = best_c[l]; }
        C.nfresh[0] = (int32_t)cnt;                          // the device keeps the cache's books from here: every column is copied in afresh
        for (int64_t t = 0; t < cnt; ++t) {
            ridx_h.p[t] = lst[(size_t)t]; rval_h.p[t] = zero_list ? 0.0 : b.val[(size_t)t];
            rslot_h.p[t] = (int32_t)t; rslot_h.p[2 * res_kcap + t] = (int32_t)t;
        }
        for (int64_t t = 0; t < cb; ++t) { ridx_h.p[2 * res_kcap + t] = best_b.idx[(size_t)t]; rval_h.p[2 * res_kcap + t] = best_b.val[(size_t)t]; }
        MIH_HIP(hipMemcpyAsync(rctl.p, &C, sizeof(C), hipMemcpyHostToDevice, s));
        if (cnt) {
            MIH_HIP(hipMemcpyAsync(ridx.p, ridx_h.p, sizeof(int64_t) * cnt, hipMemcpyHostToDevice, s));
            MIH_HIP(hipMemcpyAsync(rval.p, rval_h.p, sizeof(double) * cnt, hipMemcpyHostToDevice, s));
            MIH_HIP(hipMemcpyAsync(rslot.p, rslot_h.p, sizeof(int32_t) * cnt, hipMemcpyHostToDevice, s));
            MIH_HIP(hipMemcpyAsync(rslot.p + 2 * res_kcap, rslot_h.p + 2 * res_kcap, sizeof(int32_t) * cnt, hipMemcpyHostToDevice, s));
        }
        if (cb) {
            MIH_HIP(hipMemcpyAsync(ridx.p + 2 * res_kcap, ridx_h.p + 2 * res_kcap, sizeof(int64_t) * cb, hipMemcpyHostToDevice, s));
            MIH_HIP(hipMemcpyAsync(rval.p + 2 * res_kcap, rval_h.p + 2 * res_kcap, sizeof(double) * cb, hipMemcpyHostToDevice, s));
        }
        if (res_sharded) {            // the whole model of the iterate (every shard holds it: project_full_sharded)
            const int64_t gn = (int64_t)bg.idx.size();
            int64_t *gi = reinterpret_cast<int64_t *>(rg_h.p); double *gv = rg_h.p + res_kcap;
            for (int64_t t = 0; t < gn; ++t) { gi[t] = bg.idx[(size_t)t]; gv[t] = bg.val[(size_t)t]; }
            if (gn) {
                MIH_HIP(hipMemcpyAsync(rgidx.p, gi, sizeof(int64_t) * gn, hipMemcpyHostToDevice, s));
                MIH_HIP(hipMemcpyAsync(rgval.p, gv, sizeof(double) * gn, hipMemcpyHostToDevice, s));
            }
            const int64_t g2[2] = {gn, 0};
            rg_cnt_h[0] = g2[0]; rg_cnt_h[1] = g2[1];
            MIH_HIP(hipMemcpyAsync(rctl.p->gcnt, rg_cnt_h, sizeof(int64_t) * 2, hipMemcpyHostToDevice, s));
        }
        res_zero_list = zero_list; res_iter0 = iter_done; res_known = 0; res_fast_fails = 0; res_spec = 0; res_last[0] = res_last[1] = res_last[2] = 0;      // (no threshold on the device yet)
        res_out.clear();
        xv_cache_forget(); stage_forget(); spec_ok = false;
        res_active = true;
        MIH_TRY(res_enqueue_support());
        return MIH_OK;
    }
    // ... and back: b, c, idc, the list, df on it, df2, the best model; *next_logl / *best as fit_iht! carries them
    // handback: the device handed the step back (RES_ABORT) -- possibly after attempts it had rejected, whose sweeps overwrote xb, zc
    // and mu with those candidates' values; the host-driven replay starts with iht_stepsize!, which reads them
    int res_end(double *next_logl, double *best, bool handback = false)
    {
        if (!res_active) return MIH_OK;
        res_active = false;
        res_dead_passes((int)res_out.size());                // (steps queued ahead of the last record read: not to be run)
        res_out.clear();
        ++res_epoch;                                         // whatever is still queued does nothing
        {
            // (a one-word store through the stream: the chain's gate closes in order, behind the kernels that are running)
            const int32_t e = res_gate_step(res_epoch);
            MIH_HIP(hipMemcpyAsync(&rctl.p->live_epoch, &e, sizeof(e), hipMemcpyHostToDevice, s));
        }
        MIH_HIP(hipMemcpyAsync(rctl_h.p, rctl.p, sizeof(ResCtl), hipMemcpyDeviceToHost, s));
        MIH_HIP(hipMemcpyAsync(ridx_h.p, ridx.p, sizeof(int64_t) * 3 * res_kcap, hipMemcpyDeviceToHost, s));
        MIH_HIP(hipMemcpyAsync(rval_h.p, rval.p, sizeof(double) * 3 * res_kcap, hipMemcpyDeviceToHost, s));
        MIH_HIP(hipMemcpyAsync(hpin.p, gval.p, sizeof(double) * std::min<size_t>((size_t)res_kcap, hpin.n), hipMemcpyDeviceToHost, s));
        MIH_HIP(hipStreamSynchronize(s));
        const ResCtl &C = *rctl_h.p;
        const int cur = C.cur;
        const int64_t cnt = C.m[cur].cnt, cb = C.best.cnt;
        const int64_t *li = ridx_h.p + (size_t)cur * res_kcap; const double *lv = rval_h.p + (size_t)cur * res_kcap;
        const bool still_zero = res_zero_list && C.iter == (int32_t)res_iter0 && cur == 0;
        idx.idx.assign(li, li + cnt);
        if (still_zero) b.clear();
        else { b.idx = idx.idx; b.val.assign(lv, lv + cnt); }
        idx.val.assign(hpin.p, hpin.p + std::min<int64_t>(cnt, (int64_t)hpin.n));
        idx.val.resize((size_t)cnt, 0.0);
        for (int l = 0; l < q; ++l) { c[l] = C.m[cur].c[l]; idc[l] = (uint8_t)((C.m[cur].idc >> l) & 1ull); df2[l] = C.df2[l]; best_c[l] = C.best.c[l]; }
        best_b.idx.assign(ridx_h.p + 2 * res_kcap, ridx_h.p + 2 * res_kcap + cb);
        best_b.val.assign(rval_h.p + 2 * res_kcap, rval_h.p + 2 * res_kcap + cb);
        b0 = b; c0 = c; idc0 = idc;
        if (res_sharded) {
            const int64_t gn = C.gcnt[cur];
            int64_t *gi = reinterpret_cast<int64_t *>(rg_h.p); double *gv = rg_h.p + res_kcap;
            if (gn) {
                MIH_HIP(hipMemcpy(gi, rgidx.p + (size_t)cur * res_kcap, sizeof(int64_t) * gn, hipMemcpyDeviceToHost));
                MIH_HIP(hipMemcpy(gv, rgval.p + (size_t)cur * res_kcap, sizeof(double) * gn, hipMemcpyDeviceToHost));
            }
            bg.idx.assign(gi, gi + gn); bg.val.assign(gv, gv + gn); bg_ok = true;
            b0g = bg; b0g_ok = true;
        }
        if (next_logl) *next_logl = C.logl_cur;
        if (best) *best = C.best_logl;
        xv_cache_forget(); stage_forget(); spec_ok = false; df2_pending = false;
        if (res_lean() || handback) {  // xb, zc, mu of the iterate, which the lean attempts did not store (and the others overwrote with a rejected candidate's): k_xv_snp_cached + k_zmul + k_mu_loglik, the same sums
            MIH_TRY(update_xb());
            MIH_TRY(mu_loglik(1, nullptr, nullptr));
        }
        return MIH_OK;
    }
    int64_t res_iter0 = 0;
    int64_t rg_cnt_h[2] = {0, 0};
    // the X'r passes of the last `count` step chains were queued behind a kernel that closed the gate: they did nothing, and their
    // profile records (mih_profile_passes) go
    void res_dead_passes(int count)
    {
        Profile &pf = *h->prof;
        if (!pf.on || count <= 0) return;
        std::lock_guard<std::mutex> g(pf.mu);
        for (; count > 0 && !pf.open.empty(); --count) {
            PassRecord &r = pf.open.back();
            (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
            pf.open.pop_back();
        }
    }
    // ... of the series `behind` places before the last one queued (a series that ended without a step: its pass found phase 1)
    void res_dead_pass_at(int behind)
    {
        Profile &pf = *h->prof;
        if (!pf.on || behind < 0) return;
        std::lock_guard<std::mutex> g(pf.mu);
        if ((size_t)behind >= pf.open.size()) return;
        const size_t at = pf.open.size() - 1 - (size_t)behind;
        (void)hipEventDestroy(pf.open[at].e0); (void)hipEventDestroy(pf.open[at].e1);
        pf.open.erase(pf.open.begin() + (std::ptrdiff_t)at);
    }

    int res_enqueue_support()
    {
        const ResPtrs P = res_ptrs();
        hipLaunchKernelGGL(k_res_support, dim3(nblk(res_kcap)), dim3(256), 0, s, P, res_epoch, res_mat(), df.p);
        return MIH_OK;
    }
    // start of a step: the best model so far, X_S df_S, the step size.  Column shard: the shards' partial products and their
    // shares of |df_S|^2 (element n) meet in ONE all-reduce queued on this stream, between the product and the step-size terms.
    int res_enqueue_front()
    {
        const ResPtrs P = res_ptrs(); const ResMat M = res_mat();
        if (!res_fix() && !res_sharded)
            hipLaunchKernelGGL(k_res_xgk<false>, dim3(res_wide_blocks()), dim3(kResWideRows), kResWideLds, s, P, res_epoch, M, z.p, xb.p, zc.p, mu.p, w.p, q, dist, link, nb_r, xgk.p, red.p);
        else {
            hipLaunchKernelGGL(k_res_xgk<true>, dim3(res_wide_blocks()), dim3(kResWideRows), kResWideLds, s, P, res_epoch, M, z.p, xb.p, zc.p, mu.p, w.p, q, dist, link, nb_r, xgk.p, red.p);
            if (res_fix()) hipLaunchKernelGGL(k_res_missing, dim3(1), dim3(1024), 0, s, P, res_epoch, -1, 0, M, xgk.p);
            if (res_sharded) {
                hipLaunchKernelGGL(k_res_sumsq, dim3(1), dim3(1), 0, s, P, res_epoch, xgk.p + n);
                MIH_TRY(allreduce_dev(xgk.p, n + 1, 0));
            }
            hipLaunchKernelGGL(k_res_stepsize, dim3(nb), dim3(256), 0, s, P, res_epoch, xgk.p, z.p, xb.p, zc.p, mu.p, w.p, n, q, dist, link, nb_r, red.p);
        }
        hipLaunchKernelGGL(k_res_eta, dim3(1), dim3(256), 0, s, P, res_epoch, red.p, nb, q, res_sharded ? xgk.p + n : (const double *)nullptr);
        return MIH_OK;
    }
    // one attempt of the step (attempt 0: the first, a >= 1: the a-th backtracking one): gradient step, projection, update_xb!,
    // loglikelihood, decision.  more: the kernels of attempt a + 1 are queued right behind (a forecast; they run only if needed)
    int res_enqueue_attempt(uint64_t seq, int a, bool more, bool fast)
    {
        const ResPtrs P = res_ptrs(); const ResMat M = res_mat();
        const double *wp = has_weight ? weight.p : nullptr;
        const uint64_t zk = res_zkeep_mask();
        const int64_t len = p + q, groups = (J == 0) ? 1 : J;
        const uint64_t K = (uint64_t)(k + zkeepn);
        // column shard: the shard selects among its own p columns (its top Kloc), the covariate tail and the threshold are judged
        // over all shards in k_res_select_global, behind the all-gather of the shards' messages
        const uint64_t Ksel = res_sharded ? (uint64_t)std::min<int64_t>((int64_t)K, p) : K;
        const int64_t sel_len = res_sharded ? p : len;
        if (fast)            // the direct gather: the threshold of the last attempt `a` as forecast, verified by the select
            hipLaunchKernelGGL(k_res_grad<true>, dim3(kResGradBlocks), dim3(256), 0, s, P, res_epoch, a, df.p, wp, p, q, zk, sel_len, full.p);
        else {
            hipLaunchKernelGGL(k_res_grad<false>, dim3(kResGradBlocks), dim3(256), 0, s, P, res_epoch, a, df.p, wp, p, q, zk, sel_len, full.p);
            hipLaunchKernelGGL(k_res_hist2, dim3(kResHistBlocks), dim3(256), 0, s, P, res_epoch, a, full.p, sel_len, Ksel);
            hipLaunchKernelGGL(k_res_collect, dim3(kResCollectBlocks), dim3(256), 0, s, P, res_epoch, a, full.p, sel_len);
        }
        // (measurement build) MENDELIHT_RES_FORCE_ABORT_ES=N: the device hands a step back once it has backtracked N times -- the replay
        // of a step whose rejected attempts have been through xb, zc, mu (test_handback_after_rejected_attempts)
        static const int force_abort_es = probe_env("MENDELIHT_RES_FORCE_ABORT_ES") ? atoi(probe_env("MENDELIHT_RES_FORCE_ABORT_ES")) : -1;
        if (!res_sharded && res_big)
            hipLaunchKernelGGL(k_res_select_big, dim3(1), dim3(1024), 0, s, P, res_epoch, a, fast ? 1 : 0, K, seq, M, wp, p, q, zk, (int)zkeepn, groups * (k + zkeepn), force_abort_es);
        else if (!res_sharded)
            hipLaunchKernelGGL(k_res_select, dim3(1), dim3(1024), 0, s, P, res_epoch, a, fast ? 1 : 0, K, seq, M, wp, p, q, zk, (int)zkeepn, groups * (k + zkeepn), force_abort_es);
        else {
            hipLaunchKernelGGL(k_res_select_local, dim3(1), dim3(1024), 0, s, P, res_epoch, a, fast ? 1 : 0, Ksel, K, (int32_t)xv.slots);
            MIH_TRY(res_allgather_messages(2 + 2 * (int64_t)K));
            hipLaunchKernelGGL(k_res_select_global, dim3(1), dim3(1024), 0, s, P, res_epoch, a, K, seq, M, full.p, p, q, zk, (int)zkeepn, groups * (k + zkeepn));
        }
        if (!res_fix() && !res_sharded)
            hipLaunchKernelGGL(k_res_xb<false>, dim3(res_wide_blocks()), dim3(kResWideRows), kResWideLds, s, P, res_epoch, a, M, z.p, y.p, w.p, q, dist, link, nb_r, xb.p, zc.p, mu.p, red.p, r.p, res_lean() ? 1 : 0);
        else {
            // (the partial product goes to a scratch vector -- xgk is free here -- so that a chain whose gate is closed, whose
            // collectives run all the same, leaves xb alone)
            double *part = res_sharded ? xgk.p : xb.p;
            hipLaunchKernelGGL(k_res_xb<true>, dim3(res_wide_blocks()), dim3(kResWideRows), kResWideLds, s, P, res_epoch, a, M, z.p, y.p, w.p, q, dist, link, nb_r, part, zc.p, mu.p, red.p, r.p, 0);
            if (res_fix()) hipLaunchKernelGGL(k_res_missing, dim3(1), dim3(1024), 0, s, P, res_epoch, a, 1, M, part);
            if (res_sharded) MIH_TRY(allreduce_dev(part, n, 0));
            hipLaunchKernelGGL(k_res_mu, dim3(nb), dim3(256), 0, s, P, res_epoch, a, (const double *)part, z.p, y.p, w.p, n, q, dist, link, nb_r, xb.p, zc.p, mu.p, red.p, r.p);
        }
        hipLaunchKernelGGL(k_res_decide, dim3(1), dim3(256), 0, s, P, res_epoch, a, more ? 1 : 0, seq, red.p, nb, n, dist);
        MIH_HIP(hipGetLastError());
        return MIH_OK;
    }
    // the shards' messages of the projection (k_res_select_local -> k_res_select_global): one ncclAllGather queued on this stream
    int res_allgather_messages(int64_t mlen)
    {
        const ResPtrs P = res_ptrs();
        Profile &pf = *h->prof;
        ExchRecord rec; rec.kind = 2;
        const bool timed = pf.on && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess;
        if (timed) (void)hipEventRecord(rec.e0, s);
        const int rc = comm_native_allgather_on_stream(comm, P.msg, P.msgs, mlen, s, h->device);
        if (rc < 0) { set_error("the device-resident sharded step needs the library's own communicator"); return MIH_BAD_ARG; }
        if (timed) { (void)hipEventRecord(rec.e1, s); std::lock_guard<std::mutex> g(pf.mu); pf.xopen.push_back(rec); }
        return rc;
    }
    // A series of attempt slots: 1 + res_spec of them (res_spec: the most backtracks one of the last three steps needed, the forecast
    // for this one: a slot too many is four empty launches, one too few leaves the step's end and the next one's start empty).  A slot serves whichever attempt ctl->es says is due; a0 is only the host's guess of the first one's number.
    // res_known: attempts 0 .. res_known - 1 have a forecast on the device (a step with that many attempts has stood since
    // res_begin): their projections take the direct gather, unless the forecast has failed three times in this run of steps.
    // first_slow: the attempt being re-queued because its forecast failed.
    int res_spec = 0, res_fast_fails = 0, res_known = 0, res_last[3] = {0, 0, 0};
    static int res_spec_cap() { static const int c = probe_env("MENDELIHT_SPEC_CAP") ? atoi(probe_env("MENDELIHT_SPEC_CAP")) : 2; return c; }       // (measurement build: slots per series beyond the first)
    int res_enqueue_attempts(uint64_t seq, int a0, int max_step, bool first_slow = false)
    {
        // A lane's fit (batched): ONE slot per series and no direct gather.  Its chain runs while the other lane's fused pass holds every
        // CU, where even a slot that turns out empty costs what its ~1500 workgroups cost to schedule (each needs a CU the pass has to
        // give up), and the fits of a cross-validation are short (5-17 steps): the threshold forecast failed in 13 % of their steps
        // (configs[3]: 150 redos in 1147 steps), as many as backtracked at all.  A step that backtracks costs its fit one more round trip.
        static const bool lane_spec = probe_env("MENDELIHT_LANE_SPEC") != nullptr;          // (measurement build: the single fit's policy in the lanes)
        const bool plain = batched && !lane_spec;
        const int slots = plain ? 1 : 1 + std::max(0, std::min(max_step - std::min(a0, max_step), res_spec));      // (attempt max_step always stands: utilities.jl:484)
        for (int j = 0; j < slots; ++j) {
            const bool fast = !plain && !res_big && res_known > 0 && res_fast_fails < 3 && !(first_slow && j == 0);
            if (fast) h->prof->count(MIH_CNT_RESIDENT_DIRECT, 1);
            MIH_TRY(res_enqueue_attempt(seq, 0, j + 1 < slots, fast));
        }
        return MIH_OK;
    }
    // the score that ends the step: Z'r and the statistics of the residual the accepted attempt left, the gated X'r pass (its digit
    // kernel finishes those sums, its finalize kernel also leaves df on the new support and the coefficients of X_S df_S)
    int res_enqueue_back()
    {
        const ResPtrs P = res_ptrs();
        static_assert(kZtrBlocks == 2 * kStatBlocksRes, "k_res_stats pairs the walks of k_zt_r and k_r_stats");
        hipLaunchKernelGGL(k_res_stats, dim3(4 * kStatBlocksRes, (q + kResStatCov - 1) / kResStatCov), dim3(64), 0, s, P, res_epoch, z.p, r.p, n, q,
                           ztr.p, xtv.scal.p + xtv.rhs_cap * 4, rwalk.p, rtick.p);
        hipLaunchKernelGGL(k_res_peel, dim3(1), dim3(1024), 0, s, P, res_epoch, r.p, n, xtv.scal.p + xtv.rhs_cap * 4, xtv.peel.p);
        xtv.gate = &rctl.p->live_epoch; xtv.gate_val = res_gate_step(res_epoch); xtv.stats_done = true;
        xtv.shook.spart = xtv.scal.p + xtv.rhs_cap * 4; xtv.shook.zpart = ztr.p; xtv.shook.df2 = rctl.p->df2; xtv.shook.q = q;
        xtv.shook.zblocks = kZtrBlocks; xtv.shook.ebits = xtv.dm.ebits;
        xtv.hook.cur = &rctl.p->cur;
        for (int i = 0; i < 2; ++i) { xtv.hook.idx[i] = P.idx[i]; xtv.hook.cnt[i] = &rctl.p->m[i].cnt; }
        xtv.hook.gval = P.gval; xtv.hook.A = P.coefA; xtv.hook.B = P.coefB; xtv.hook.blocks = (int)nblk(res_kcap);
        const int rc = xtv_device(h, xtv, r.p, 1, df.p, s);
        xtv.gate = nullptr; xtv.gate_val = 0; xtv.stats_done = false; xtv.hook = XtvSupportHook(); xtv.shook = XtvStatsHook();
        return rc;
    }
    // The record of step chain `seq`.  No event sits in the stream for it (an event record is a queue operation of its own, several us
    // between two kernels of the chain): the host polls the pinned ring -- briefly, then with short sleeps (it runs a step ahead of
    // the records it reads, and an X'r pass of tens of ms is in front of most of them), looking at the stream now and then so that a
    // failed launch does not leave it waiting.
    // `on`: the stream the chain was queued on (a lane's batched series run on the LANE's stream, not on this fit's)
    int res_wait(uint64_t seq, ResRecord *out, hipStream_t on = nullptr)
    {
        const hipStream_t sq = on ? on : s;
        volatile ResRecord *slot_ = rrec.p + (seq % kResRing);
        const auto t0 = std::chrono::steady_clock::now();
        auto next_query = t0 + std::chrono::milliseconds(50);
        for (unsigned it = 0;; ++it) {
            if (__atomic_load_n(&slot_->seq, __ATOMIC_ACQUIRE) == seq) break;
            if (coop_can_yield()) { current_coop()->yield(); continue; }
            __builtin_ia32_pause();
            if ((it & 255u) != 255u) continue;
            const auto now = std::chrono::steady_clock::now();
            if (now - t0 > std::chrono::microseconds(200)) std::this_thread::sleep_for(std::chrono::microseconds(100));
            if (now > next_query) {
                next_query = now + std::chrono::milliseconds(50);
                const hipError_t e = hipStreamQuery(sq);
                if (e == hipSuccess) {                     // everything queued has run: the record is there, or never will be
                    if (__atomic_load_n(&slot_->seq, __ATOMIC_ACQUIRE) == seq) break;
                    set_error("device-resident step %llu left no record", (unsigned long long)seq);
                    return MIH_HIP_ERROR;
                }
                if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery", __FILE__, __LINE__);
            }
        }
        *out = *const_cast<ResRecord *>(slot_);
        return MIH_OK;
    }
    // One accepted step of the resident chain (the chain itself runs up to two steps ahead of this call, never past rr.limit
    // steps in all).  *aborted: the device met a case it leaves to the host-driven step (ties for _choose!, lists beyond its
    // buffers): the iterate is back on the host as it was when that step began, nothing of it has been applied.
    int res_next(ResRun &rr, ResRecord *rec, bool *aborted)
    {
        *aborted = false;
        for (;;) {
            while (rr.issued < rr.limit && res_out.size() < 2) {
                const uint64_t seq = ++res_seq;
                MIH_TRY(res_enqueue_front()); MIH_TRY(res_enqueue_attempts(seq, 0, rr.max_step)); MIH_TRY(res_enqueue_back());
                res_out.push_back(seq); ++rr.issued;
            }
            if (res_out.empty()) { set_error("no device-resident step in flight"); return MIH_BAD_ARG; }
            const uint64_t seq = res_out.front();
            res_out.erase(res_out.begin());
            MIH_TRY(res_wait(seq, rec));
            switch (rec->status) {
            case RES_ACCEPT:
                ++rr.done; res_known = std::max(res_known, rec->nbt + 1);
                res_last[2] = res_last[1]; res_last[1] = res_last[0]; res_last[0] = rec->nbt;
                res_spec = std::min(res_spec_cap(), std::max(res_last[0], std::max(res_last[1], res_last[2])));
                h->prof->count(MIH_CNT_RESIDENT_STEPS, 1);
                return MIH_OK;
            case RES_PENDING:                // the series ended with the step still backtracking: the series queued behind goes on with it
                res_dead_pass_at((int)res_out.size());       // (the step-end kernels in between, the X'r pass among them, did nothing)
                --rr.issued;
                h->prof->count(MIH_CNT_RESIDENT_ATTEMPTS, 1);
                break;
            case RES_REDO_SLOW: {            // the direct gather's forecast failed: the same attempt again, with the histograms
                res_dead_passes(1 + (int)res_out.size());
                ++res_epoch; res_out.clear(); rr.issued = rr.done;
                ++res_fast_fails;
                h->prof->count(MIH_CNT_RESIDENT_REDOS, 1);
                const uint64_t s2 = ++res_seq;
                MIH_TRY(res_enqueue_attempts(s2, rec->nbt, rr.max_step, true)); MIH_TRY(res_enqueue_back());
                res_out.push_back(s2); ++rr.issued;
                break;
            }
            case RES_STOP_CONVERGED: res_dead_passes(1 + (int)res_out.size()); ++res_epoch; res_out.clear(); ++rr.done; rr.issued = rr.done; h->prof->count(MIH_CNT_RESIDENT_STEPS, 1); return MIH_OK;
            case RES_STOP_NAN: case RES_STOP_INF: res_dead_passes(1 + (int)res_out.size()); ++res_epoch; res_out.clear(); ++rr.done; rr.issued = rr.done; return MIH_OK;
            case RES_ABORT:
                res_dead_passes(1 + (int)res_out.size());
                res_out.clear(); rr.issued = rr.done;      // (res_end moves the epoch on)
                h->prof->count(MIH_CNT_RESIDENT_HANDBACKS, 1);
                *aborted = true;
                return MIH_OK;
            default: set_error("device-resident step: unknown record status %d", rec->status); return MIH_HIP_ERROR;
            }
        }
    }

    // ---- a lock-step lane's fit, resident (round 6; cross_validation.jl:100-121, fit.jl:213-263) --------------------------------------
    // The lane scores the residuals of all its fits with ONE fused pass per round, so a fit's chain has no step-end of its own: behind
    // the pass the fit queues Z'r (into the control block), df on its support, the step's start and its attempt slots -- WITHOUT
    // waiting (lane_queue_step) -- and reads the step's record when the lane collects the residuals of its next pass
    // (lane_collect_step): one host wait per step and fit instead of three plus two per backtrack, and the chains of a lane's fits
    // run on the device while its host thread is still queuing the others'.
    uint64_t lane_seq = 0; bool lane_queued = false;
    int lane_queue_step(double next_logl, double best, int64_t iter_done, const mih_fit_params *prm)
    {
        lane_queued = false;
        if (!res_ok) return MIH_OK;
        if (!res_active) {               // (the fit's first step, or the step after one the device handed back)
            if (res_begin(next_logl, best, iter_done, 1, prm) != MIH_OK) { res_ok = false; return MIH_OK; }
        } else {
            hipLaunchKernelGGL(k_zt_r, dim3(kZtrBlocks, q), dim3(256), 0, s, z.p, r.p, n, ztr.p, ztr_done.p, rctl.p->df2);
            MIH_TRY(res_enqueue_support());
        }
        lane_seq = ++res_seq;
        MIH_TRY(res_enqueue_front());
        MIH_TRY(res_enqueue_attempts(lane_seq, 0, prm->max_step));
        lane_queued = true;
        return MIH_OK;
    }
    // *stepped = false: no chain was queued, or the device handed the step back (the iterate is home again): the host-driven step_pre
    // follows.  *stop: the device applied the stopping rule (fit.jl:197) and it held.
    int lane_collect_step(const mih_fit_params *prm, double *next_logl, double *best, int *nbt, double *tol, bool *stepped, bool *stop)
    {
        *stepped = false; *stop = false;
        if (!lane_queued) return MIH_OK;
        lane_queued = false;
        uint64_t seq = lane_seq;
        for (;;) {
            ResRecord rec;
            MIH_TRY(res_wait(seq, &rec));
            switch (rec.status) {
            case RES_ACCEPT: case RES_STOP_CONVERGED:
                res_known = std::max(res_known, rec.nbt + 1);
                res_last[2] = res_last[1]; res_last[1] = res_last[0]; res_last[0] = rec.nbt;
                res_spec = std::min(res_spec_cap(), std::max(res_last[0], std::max(res_last[1], res_last[2])));
                h->prof->count(MIH_CNT_RESIDENT_STEPS, 1);
                if (rec.status == RES_STOP_CONVERGED) { ++res_epoch; *stop = true; }
                *next_logl = rec.logl; *nbt = rec.nbt; *tol = rec.tol; *stepped = true;
                return MIH_OK;
            case RES_PENDING:                // the series ended with the step still backtracking: another series of slots goes on with it
                h->prof->count(MIH_CNT_RESIDENT_ATTEMPTS, 1);
                seq = ++res_seq;
                MIH_TRY(res_enqueue_attempts(seq, rec.nbt, prm->max_step));
                break;
            case RES_REDO_SLOW:              // the direct gather's forecast failed: the same attempt again, with the histograms
                ++res_epoch; ++res_fast_fails;
                h->prof->count(MIH_CNT_RESIDENT_REDOS, 1);
                seq = ++res_seq;
                MIH_TRY(res_enqueue_attempts(seq, rec.nbt, prm->max_step, true));
                break;
            case RES_STOP_NAN: case RES_STOP_INF:
                ++res_epoch;
                MIH_TRY(res_end(nullptr, nullptr));
                if (rec.status == RES_STOP_NAN) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
                set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL;
            case RES_ABORT:
                h->prof->count(MIH_CNT_RESIDENT_HANDBACKS, 1);
                return res_end(next_logl, best, true);
            default: set_error("device-resident step: unknown record status %d", rec.status); return MIH_HIP_ERROR;
            }
        }
    }

    // ... and BATCHED over the lane's fits (k_lane_*, resident.inc): this fit's record of the array the lane's kernels read.
    // step_start: the step begins here (X_S df_S, the step size); new_score: Z'r and df on the support are taken from the score that
    // has just arrived (not for a fit res_begin has just set up: it brought them along)
    void lane_fill(LaneFit &a, uint64_t seq, bool step_start, bool new_score) const
    {
        static const int force_abort_es = probe_env("MENDELIHT_RES_FORCE_ABORT_ES") ? atoi(probe_env("MENDELIHT_RES_FORCE_ABORT_ES")) : -1;
        a.P = res_ptrs(); a.M = res_mat(); a.epoch = res_epoch; a.front = step_start ? 1 : 0; a.score = new_score ? 1 : 0;
        a.q = q; a.dist = dist; a.link = link; a.zkeepn = (int)zkeepn; a.lean = res_lean() ? 1 : 0; a.force_abort_es = force_abort_es;
        a.nb_r = nb_r; a.zkeep = res_zkeep_mask(); a.K = (uint64_t)(k + zkeepn); a.seq = seq;
        a.max_nonzero = ((J == 0) ? 1 : J) * (k + zkeepn); a.p = p;
        a.z = z.p; a.y = y.p; a.w = w.p; a.weight = has_weight ? weight.p : nullptr;
        a.xb = xb.p; a.zc = zc.p; a.mu = mu.p; a.r = r.p; a.xgk = xgk.p; a.red = red.p; a.df = df.p; a.full = full.p;
        a.ztr = ztr.p; a.df2 = rctl.p->df2; a.ztr_done = ztr_done.p; a.nb = nb; a.pad = 0;
    }
    // what a record of the batched chain means for this fit (the cases of lane_collect_step).  *again: the step is still backtracking,
    // the lane queues another series for it
    int lane_take_record(const ResRecord &rec, double *next_logl, double *best, int *nbt, double *tol, bool *stepped, bool *again)
    {
        *stepped = false; *again = false;
        switch (rec.status) {
        case RES_ACCEPT: case RES_STOP_CONVERGED:
            h->prof->count(MIH_CNT_RESIDENT_STEPS, 1);
            if (rec.status == RES_STOP_CONVERGED) ++res_epoch;
            *next_logl = rec.logl; *nbt = rec.nbt; *tol = rec.tol; *stepped = true;
            return MIH_OK;
        case RES_PENDING: h->prof->count(MIH_CNT_RESIDENT_ATTEMPTS, 1); *again = true; return MIH_OK;
        case RES_STOP_NAN: case RES_STOP_INF:
            ++res_epoch;
            MIH_TRY(res_end(nullptr, nullptr));
            if (rec.status == RES_STOP_NAN) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
            set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL;
        case RES_ABORT:
            h->prof->count(MIH_CNT_RESIDENT_HANDBACKS, 1);
            return res_end(next_logl, best, true);
        default: set_error("device-resident step: unknown record status %d", rec.status); return MIH_HIP_ERROR;
        }
    }

    // fit_iht! (fit.jl:145-207)
    int fit_loop(const mih_fit_params *prm, double *best_out, int64_t *iter_out, double *lt, double *tt,
                 int32_t *btt, int32_t *ntrace)
    {
        double next_logl = -std::numeric_limits<double>::infinity(), best = next_logl;
        int64_t mm = 0; int32_t nt = 0;
        ResRun rr; rr.limit = std::max<int64_t>(0, (int64_t)prm->max_iter - 1); rr.max_step = prm->max_step;       // fit.jl:170: max_iter = N performs N - 1 steps
        auto finish = [&](int iter) -> int {
            if (res_active) MIH_TRY(res_end(&next_logl, &best));
            best = save_prev(next_logl, best);
            MIH_TRY(save_best_model());
            mm = iter;
            return MIH_OK;
        };
        for (int iter = 1; iter <= prm->max_iter; ++iter) {
            if (iter >= prm->max_iter) { MIH_TRY(finish(iter)); break; }
            int nbt = 0; double sc = 0.0;
            bool stepped = false;
            if (res_ok && !res_active) {                 // (first step, or the step after one the device handed back)
                if (res_begin(next_logl, best, iter - 1, 1, prm) == MIH_OK) rr.issued = rr.done = iter - 1;
                else res_ok = false;
            }
            if (res_active) {
                ResRecord rec; bool aborted = false;
                MIH_TRY(res_next(rr, &rec, &aborted));
                if (aborted) MIH_TRY(res_end(&next_logl, &best, true));
                else {
                    if (rec.status == RES_STOP_NAN || rec.status == RES_STOP_INF) {
                        MIH_TRY(res_end(nullptr, nullptr));
                        if (rec.status == RES_STOP_NAN) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
                        set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL;
                    }
                    next_logl = rec.logl; nbt = rec.nbt; sc = rec.tol; stepped = true;
                }
            }
            if (!stepped) {
                // the host-driven step.  debias! (fit.jl:188) and the convergence test (fit.jl:197) read b, b0, c, c0 only, so they run
                // in front of the X'r pass that ends the step; a fit that converges here skips that pass (the reference computes the
                // score inside iht_one_step! and never reads it) -- as the device-resident chain and the lock-step drivers do
                best = save_prev(next_logl, best);
                MIH_TRY(step_pre(next_logl, prm->max_step, &nbt, &next_logl));
                if (debias && iter >= 5 && comm) MIH_TRY(debias_sharded());
                else if (debias && iter >= 5 && b.idx == b0.idx && !b.idx.empty())      // fit.jl:188: v.idx == v.idx0 && debias!(v)
                    MIH_TRY(debias_glm_device(h, b.idx.data(), (int64_t)b.idx.size(), y.p, dist, link, nb_r, b.val.data(), s));
                sc = check_convergence();
                if (iter >= prm->min_iter && sc < prm->tol) {
                    if (std::isnan(next_logl)) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
                    if (std::isinf(next_logl)) { set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL; }
                } else {
                    MIH_TRY(xtv_device(h, xtv, r.p, 1, df.p, s));
                    MIH_TRY(step_post(next_logl));
                }
            }
            if (lt) lt[nt] = next_logl;
            if (tt) tt[nt] = sc;
            if (btt) btt[nt] = nbt;
            nt++;
            if (prm->progress) prm->progress(prm->progress_user, iter, next_logl, nbt, sc);
            if (iter >= prm->min_iter && sc < prm->tol) { MIH_TRY(finish(iter)); break; }
        }
        if (res_active) MIH_TRY(res_end(&next_logl, &best));       // (max_iter = 0: no iteration at all)
        *best_out = best; *iter_out = mm;
        if (ntrace) *ntrace = nt;
        return MIH_OK;
    }
};

static int check_params(const mih_mat *h, const mih_fit_params *prm, int64_t q)
{
    if (!h || !prm) { set_error("null handle/params"); return MIH_BAD_ARG; }
    if (prm->J < 0) { set_error("Value of J (max number of groups) must be nonnegative!"); return MIH_BAD_ARG; }
    if (prm->max_iter < 0) { set_error("Value of max_iter must be nonnegative!"); return MIH_BAD_ARG; }
    if (prm->max_step < 0) { set_error("Value of max_step must be nonnegative!"); return MIH_BAD_ARG; }
    if (prm->cv_threads < 0) { set_error("cv_threads must be nonnegative (0 = 1 = one chain, the reference at Threads.nthreads() == 1)"); return MIH_BAD_ARG; }
    if (!(prm->tol > 2.220446049250313e-16)) { set_error("Value of global tol must exceed machine precision!"); return MIH_BAD_ARG; }
    if (h->kind == 0 && !h->center) { set_error("x is not centered! Please construct SnpLinAlg{Float64}(::SnpArray, center=true, scale=true)"); return MIH_NOT_CENTERED; }
    if (prm->est_r != MIH_ESTR_NONE && prm->dist != MIH_NEGBIN) { set_error("Only negative binomial regression currently supports nuisance parameter estimation"); return MIH_BAD_ARG; }
    if (!prm->ks && prm->k < 0) { set_error("Value of k (max predictors per group) must be nonnegative!"); return MIH_BAD_ARG; }
    if (prm->ks) for (int64_t g = 0; g < prm->nks; ++g)                 // fit.jl:87 for a vector k: the same bound on every entry
        if (prm->ks[g] < 0) { set_error("Value of k (max predictors per group) must be nonnegative!"); return MIH_BAD_ARG; }
    if (q < 1 || q > kMaxQ) { set_error("number of covariates q=%lld must be in 1..%d", (long long)q, kMaxQ); return MIH_BAD_DIM; }
    if (prm->dist < 0 || prm->dist > MIH_INVGAUSS || prm->link < 0 || prm->link > MIH_SQRT) { set_error("unknown distribution/link"); return MIH_BAD_ARG; }
    return MIH_OK;
}


}  // namespace mih
