// debias.hip -- debias! (src/utilities.jl:1014-1020): after an IHT step that kept the support, refit the
// support columns by a GLM and overwrite b[idx] with its coefficients:
//     temp_glm = fit(GeneralizedLinearModel, v.xk, v.y, v.d, v.l);  view(v.b, v.idx) .= temp_glm.pp.beta0
// The model has no intercept and no covariates, uses ALL n samples with unit weights (cv_wts is not
// passed), and is fitted by GLM.jl's IRLS (glmfit.jl `_fit!`; GLM.jl is a dependency, not vendored in the reference):
// mustart -> first weighted least squares on the working response -> delbeta!/step halving until
// devold - dev < max(rtol*devold, atol) (rtol = atol = 1e-6, <= 30 steps, minstepfac 1e-3).
//
// The reference needs its dense n x k copy `xk` for this (memory_efficient=false); here the k support
// columns are decoded from the 2-bit matrix into an n x (k+1) f64 panel in HBM (the extra column holds
// the working response), X'WX and X'Wr come from one tiled Gram kernel over the panel (fixed-order
// two-stage sums, no atomics), and the k x k Cholesky solve runs on the host like the r x r algebra of
// the multivariate path.
#include "common.h"
#include "fit_common.h"
#include <cmath>
#include <cstdio>
#include <limits>
#include <vector>

namespace mih {

// panel[i + n*t] = standardized x[i, idx[t]] with missing entries as stored (dosage 0); fixed below
__global__ void __launch_bounds__(256)
k_db_cols_snp(const uint32_t *__restrict__ X, int64_t nbp, int64_t ndw, int64_t n, const int64_t *__restrict__ idx,
              const double *__restrict__ mu, const double *__restrict__ sinv, int center, int scale,
              double *__restrict__ panel)
{
    int64_t dw = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (dw >= ndw) return;
    const int64_t t = blockIdx.y, j = idx[t];
    const double s = scale ? sinv[j] : 1.0, cm = center ? mu[j] : 0.0;
    const uint32_t w = X[xword(nbp, j, dw)];
    #pragma unroll
    for (int r = 0; r < 16; ++r) {
        int64_t i = dw * 16 + r;
        if (i < n) panel[i + n * t] = ((double)((w >> (2 * r)) & 3u) - cm) * s;
    }
}
__global__ void k_db_cols_missing(const int64_t *__restrict__ idx, const double *__restrict__ mu, const double *__restrict__ sinv,
                                  int center, int scale, int impute, const int64_t *__restrict__ miss_ptr,
                                  const int32_t *__restrict__ miss_row, int64_t n, double *__restrict__ panel)
{
    const int64_t t = blockIdx.x, j = idx[t];
    const double s = scale ? sinv[j] : 1.0, cm = center ? mu[j] : 0.0;
    const double v = ((impute ? mu[j] : 0.0) - cm) * s;
    for (int64_t e = miss_ptr[j] + threadIdx.x; e < miss_ptr[j + 1]; e += blockDim.x) panel[miss_row[e] + n * t] = v;
}
template <typename T>
__global__ void k_db_cols_dense(const T *__restrict__ D, int64_t n, const int64_t *__restrict__ idx, double *__restrict__ panel)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) panel[i + n * blockIdx.y] = (double)D[idx[blockIdx.y] * n + i];
}

// GlmResp initialisation: mu = mustart(y), eta = linkfun(mu)
__global__ void k_db_start(const double *__restrict__ y, int64_t n, int dist, int link, double *__restrict__ eta)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) eta[i] = d_linkfun(link, d_mustart(dist, y[i]));
}
// updateMu!: eta -> working residual (y-mu)/mueta, working weight mueta^2/var, deviance partial sums;
// target = wrkresid (+ eta for the first solve on the working response) goes into the panel's last column
__global__ void __launch_bounds__(256)
k_db_update(const double *__restrict__ eta, const double *__restrict__ y, int64_t n, int dist, int link, double nb_r,
            int add_eta, double *__restrict__ target, double *__restrict__ wwt, double *__restrict__ part)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    if (i < n) {
        double e = eta[i], m = d_linkinv(link, e), me = d_mueta(link, e);
        double wr = (y[i] - m) / me;
        target[i] = add_eta ? e + wr : wr;
        wwt[i] = me * me / d_glmvar(dist, m, nb_r);
        v[0] = d_devresid(dist, y[i], m, nb_r);
    }
    block_sum<1>(v, part + blockIdx.x);
}
// lp = panel[:, 0:k] * coef
__global__ void __launch_bounds__(256)
k_db_linpred(const double *__restrict__ panel, int64_t n, int k, const double *__restrict__ coef, double *__restrict__ lp)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    for (int t = 0; t < k; ++t) a += panel[i + n * t] * coef[t];
    lp[i] = a;
}

// Gram matrix of the panel's m = k+1 columns with weights w: G[a][b] = sum_i P[i,a] w_i P[i,b] for the
// upper-triangular 16x16 tiles.  grid (tile pair, row slice); per-slice partial tiles are summed in slice
// order by k_db_gram_reduce.
constexpr int kGramTile = 16, kGramSlices = 64, kGramChunk = 64;
__global__ void __launch_bounds__(256)
k_db_gram(const double *__restrict__ P, int64_t n, int m, const double *__restrict__ w, const int32_t *__restrict__ tiles,
          double *__restrict__ part /* [pair][slice][16][16] */)
{
    __shared__ double sa[kGramChunk][kGramTile + 1], sb[kGramChunk][kGramTile + 1];
    const int ta = tiles[2 * blockIdx.x], tb = tiles[2 * blockIdx.x + 1];
    const int a = threadIdx.x / kGramTile, b = threadIdx.x % kGramTile;
    const int64_t per = (n + kGramSlices - 1) / kGramSlices;
    const int64_t r0 = blockIdx.y * per, r1 = (r0 + per < n) ? r0 + per : n;
    double acc = 0.0;
    for (int64_t base = r0; base < r1; base += kGramChunk) {
        // stage kGramChunk rows x 16 columns of both column tiles (column-major panel: coalesced along rows)
        for (int e = threadIdx.x; e < kGramChunk * kGramTile; e += 256) {
            int c = e / kGramChunk, r = e % kGramChunk;
            int64_t i = base + r;
            int ca = ta * kGramTile + c, cb = tb * kGramTile + c;
            double wa = (i < r1 && ca < m) ? P[i + n * ca] * w[i] : 0.0;
            double xb = (i < r1 && cb < m) ? P[i + n * cb] : 0.0;
            sa[r][c] = wa; sb[r][c] = xb;
        }
        __syncthreads();
        #pragma unroll 8
        for (int r = 0; r < kGramChunk; ++r) acc = fma(sa[r][a], sb[r][b], acc);
        __syncthreads();
    }
    part[((int64_t)blockIdx.x * kGramSlices + blockIdx.y) * (kGramTile * kGramTile) + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256)
k_db_gram_reduce(const double *__restrict__ part, const int32_t *__restrict__ tiles, int m, double *__restrict__ G /* m x m, upper */)
{
    const int ta = tiles[2 * blockIdx.x], tb = tiles[2 * blockIdx.x + 1];
    const int a = threadIdx.x / kGramTile, b = threadIdx.x % kGramTile;
    double s = 0.0;
    for (int sl = 0; sl < kGramSlices; ++sl) s += part[((int64_t)blockIdx.x * kGramSlices + sl) * (kGramTile * kGramTile) + threadIdx.x];
    int ra = ta * kGramTile + a, cb = tb * kGramTile + b;
    if (ra < m && cb < m) G[ra + (int64_t)m * cb] = s;
}

// unpivoted Cholesky solve of the k x k system held in the upper triangle of G (leading dimension m)
static bool chol_solve_upper(std::vector<double> &G, int m, int k, std::vector<double> &x)
{
    for (int j = 0; j < k; ++j) {
        double d = G[j + (size_t)m * j];
        for (int l = 0; l < j; ++l) d -= G[l + (size_t)m * j] * G[l + (size_t)m * j];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d); G[j + (size_t)m * j] = d;
        for (int i = j + 1; i < k; ++i) {
            double s = G[j + (size_t)m * i];
            for (int l = 0; l < j; ++l) s -= G[l + (size_t)m * j] * G[l + (size_t)m * i];
            G[j + (size_t)m * i] = s / d;
        }
    }
    for (int j = 0; j < k; ++j) {
        double s = x[j];
        for (int l = 0; l < j; ++l) s -= G[l + (size_t)m * j] * x[l];
        x[j] = s / G[j + (size_t)m * j];
    }
    for (int j = k - 1; j >= 0; --j) {
        double s = x[j];
        for (int l = j + 1; l < k; ++l) s -= G[j + (size_t)m * l] * x[l];
        x[j] = s / G[j + (size_t)m * j];
    }
    return true;
}

int debias_glm_device(const mih_mat *h, const int64_t *idx_host, int64_t k64, const double *y_dev, int dist, int link,
                      double nb_r, double *beta_out, hipStream_t s, const DebiasShard *shard)
{
    const int64_t n = h->n;
    const int kloc = (int)k64;                                  // the columns this process decodes (all of them without a shard)
    const int k = shard ? (int)shard->k_total : kloc, m = k + 1;
    const int koff = shard ? (int)shard->k_off : 0;
    if (k == 0) return MIH_OK;
    if (link == MIH_PROBIT) { set_error("debias is not available with ProbitLink (no closed-form link function)"); return MIH_BAD_ARG; }
    const int T = (m + kGramTile - 1) / kGramTile;
    std::vector<int32_t> tiles;
    for (int a = 0; a < T; ++a) for (int b = a; b < T; ++b) { tiles.push_back(a); tiles.push_back(b); }
    const int npairs = (int)tiles.size() / 2, nb = (int)nblk(n);
    DevBuf<double> panel, eta, wwt, coef, part, G, red, scal; DevBuf<int64_t> idx; DevBuf<int32_t> tl;
    MIH_TRY(panel.alloc((size_t)n * m)); MIH_TRY(eta.alloc(n)); MIH_TRY(wwt.alloc(n)); MIH_TRY(coef.alloc(k));
    MIH_TRY(part.alloc((size_t)npairs * kGramSlices * kGramTile * kGramTile)); MIH_TRY(G.alloc((size_t)m * m));
    MIH_TRY(red.alloc(nb)); MIH_TRY(scal.alloc(4)); MIH_TRY(idx.alloc(kloc > 0 ? kloc : 1)); MIH_TRY(tl.alloc(tiles.size()));
    if (kloc > 0) MIH_HIP(hipMemcpyAsync(idx.p, idx_host, sizeof(int64_t) * kloc, hipMemcpyHostToDevice, s));
    MIH_HIP(hipMemcpyAsync(tl.p, tiles.data(), sizeof(int32_t) * tiles.size(), hipMemcpyHostToDevice, s));
    if (shard) MIH_HIP(hipMemsetAsync(panel.p, 0, sizeof(double) * (size_t)n * k, s));       // (the other shards' columns: zeros, until the sum)
    double *mine = panel.p + (size_t)n * koff;                  // this process's columns of the panel
    if (kloc > 0 && h->kind == 0) {
        int64_t ndw = h->n_pad / 16;
        hipLaunchKernelGGL(k_db_cols_snp, dim3(nblk(ndw), (unsigned)kloc), dim3(256), 0, s, h->X, h->nbp, ndw, n, idx.p, h->mu, h->sinv,
                           h->center, h->scale, mine);
        if (h->total_missing > 0)
            hipLaunchKernelGGL(k_db_cols_missing, dim3((unsigned)kloc), dim3(256), 0, s, idx.p, h->mu, h->sinv, h->center, h->scale,
                               h->impute, h->miss_ptr, h->miss_row, n, mine);
    } else if (kloc > 0) {
        if (h->Df) hipLaunchKernelGGL(k_db_cols_dense<float>, dim3(nblk(n), (unsigned)kloc), dim3(256), 0, s, h->Df, n, idx.p, mine);
        else hipLaunchKernelGGL(k_db_cols_dense<double>, dim3(nblk(n), (unsigned)kloc), dim3(256), 0, s, h->D, n, idx.p, mine);
    }
    if (shard) MIH_TRY(shard->reduce(panel.p, n * (int64_t)k));
    double *target = panel.p + (size_t)n * k;        // last panel column: the right-hand side of the WLS

    auto update = [&](int add_eta, double *dev) -> int {      // updateMu! + deviance
        hipLaunchKernelGGL(k_db_update, dim3(nb), dim3(256), 0, s, eta.p, y_dev, n, dist, link, nb_r, add_eta, target, wwt.p, red.p);
        hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, s, red.p, nb, 1, scal.p);
        MIH_HIP(hipMemcpyAsync(dev, scal.p, sizeof(double), hipMemcpyDeviceToHost, s));
        MIH_HIP(hipStreamSynchronize(s));
        if (std::isnan(*dev)) *dev = std::numeric_limits<double>::infinity();
        return MIH_OK;
    };
    std::vector<double> Gh((size_t)m * m), del(k), beta0(k, 0.0), trial(k);
    auto delbeta = [&]() -> int {                             // delbeta!: (X'WX) del = X'W target
        hipLaunchKernelGGL(k_db_gram, dim3((unsigned)npairs, kGramSlices), dim3(256), 0, s, panel.p, n, m, wwt.p, tl.p, part.p);
        hipLaunchKernelGGL(k_db_gram_reduce, dim3((unsigned)npairs), dim3(256), 0, s, part.p, tl.p, m, G.p);
        MIH_HIP(hipMemcpyAsync(Gh.data(), G.p, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToHost, s));
        MIH_HIP(hipStreamSynchronize(s));
        for (int a = 0; a < k; ++a) del[a] = Gh[a + (size_t)m * k];
        if (!chol_solve_upper(Gh, m, k, del)) { set_error("debias: X'WX of the support columns is not positive definite"); return MIH_BAD_ARG; }
        return MIH_OK;
    };
    auto linpred = [&](double f) -> int {                     // eta = X (beta0 + f del)
        for (int a = 0; a < k; ++a) trial[a] = beta0[a] + f * del[a];
        MIH_HIP(hipMemcpyAsync(coef.p, trial.data(), sizeof(double) * k, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_db_linpred, dim3(nb), dim3(256), 0, s, panel.p, n, k, coef.p, eta.p);
        return MIH_OK;
    };

    static const bool trace = probe_env("MENDELIHT_DEBIAS_TRACE") != nullptr;      // (measurement build: the IRLS iterates, tools/repro_fuzz.py)
    double dev, devold;
    hipLaunchKernelGGL(k_db_start, dim3(nb), dim3(256), 0, s, y_dev, n, dist, link, eta.p);
    MIH_TRY(update(1, &dev));                                 // wrkresp = eta + wrkresid at mustart
    MIH_TRY(delbeta());
    MIH_TRY(linpred(1.0));
    beta0 = trial;                                            // installbeta! from beta0 = 0
    MIH_TRY(update(0, &devold));
    if (trace) fprintf(stderr, "debias: k %d start dev %.15g\n", k, devold);
    bool cvg = false;
    for (int it = 1; it <= 30; ++it) {
        MIH_TRY(delbeta());
        double f = 1.0;
        MIH_TRY(linpred(f));
        MIH_TRY(update(0, &dev));
        while (dev > devold + 1e-6 * dev) {                   // step halving
            f /= 2.0;
            if (!(f > 0.001)) { set_error("debias: step-halving failed"); return MIH_BAD_ARG; }
            MIH_TRY(linpred(f));
            MIH_TRY(update(0, &dev));
        }
        beta0 = trial;
        if (trace) fprintf(stderr, "debias: it %d f %g dev %.15g devold-dev %.3e\n", it, f, dev, devold - dev);
        if (devold - dev < std::fmax(1e-6 * devold, 1e-6)) { cvg = true; break; }
        if (!std::isfinite(dev)) { set_error("debias: non-finite deviance"); return MIH_BAD_ARG; }
        devold = dev;
    }
    if (!cvg) { set_error("debias: the GLM refit did not converge in 30 iterations"); return MIH_BAD_ARG; }
    for (int a = 0; a < k; ++a) beta_out[a] = beta0[a];
    return MIH_OK;
}

}  // namespace mih
