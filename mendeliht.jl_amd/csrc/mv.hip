// mv.hip -- multivariate-Gaussian IHT (src/multivariate.jl) on the device.
//
// Layout: every r x n matrix of the reference (Y, BX, CZ, mu, resid, r_by_n1) is kept as r
// "planes" of n doubles (trait-major), which is exactly the n x r column-major operand the
// batched X'R pass takes (multivariate.jl:83-86: n_by_r = r_by_n1', p_by_r = X * n_by_r).
// df is r planes of p doubles.  B is k-sparse and lives on the host as (linear index i + r*j,
// value) pairs in the reference's vec(B) order; C, df2, Gamma are small host matrices
// (column-major r x q / r x r, as in Julia).  The r x r algebra (pivoted Cholesky, inverse,
// logdet) runs on the host exactly as the reference leaves it to LAPACK.
#include "common.h"
#include "fit_common.h"
#include <map>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <memory>

namespace mih {

constexpr int kMaxR = 32;
constexpr int kMaxRQ = 256;               // r * q entries of C passed by value
struct RMat { double v[kMaxR * kMaxR]; }; // column-major r x r
struct CMat { double v[kMaxRQ]; };        // column-major r x q
constexpr int kRedBlocks = 128;

// mu = BX + CZ (multivariate.jl:39-43), CZ = C*Z (:30), resid = (Y - mu) * cv_wts (:50-58)
__global__ void __launch_bounds__(256)
k_mv_resid(const double *__restrict__ Y, const double *__restrict__ Z, const double *__restrict__ BX,
           const double *__restrict__ w, int64_t n, int r, int q, CMat C, double *__restrict__ MU,
           double *__restrict__ RES)
{
    int64_t s = blockIdx.x * 256ll + threadIdx.x;
    if (s >= n) return;
    double wt = w[s];
    for (int i = 0; i < r; ++i) {
        double cz = 0.0;
        for (int l = 0; l < q; ++l) cz += C.v[i + r * l] * Z[(int64_t)l * n + s];
        double m = BX[(int64_t)i * n + s] + cz;
        MU[(int64_t)i * n + s] = m;
        RES[(int64_t)i * n + s] = (Y[(int64_t)i * n + s] - m) * wt;
    }
}

// dot products of plane pairs: pair t = (a[t], b[t]) -> part[t][block]
__global__ void __launch_bounds__(256)
k_mv_dots(const double *__restrict__ A, const double *__restrict__ B, int64_t n, const int32_t *__restrict__ pa,
          const int32_t *__restrict__ pb, double *__restrict__ part)
{
    const double *x = A + (int64_t)pa[blockIdx.y] * n, *y = B + (int64_t)pb[blockIdx.y] * n;
    double v[1] = {0.0};
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * kRedBlocks) v[0] += x[i] * y[i];
    block_sum<1>(v, part + (int64_t)blockIdx.y * kRedBlocks + blockIdx.x);
}
__global__ void __launch_bounds__(64)
k_mv_dots_final(const double *__restrict__ part, double *__restrict__ out)
{
    if (threadIdx.x != 0) return;
    double a = 0.0;
    for (int b = 0; b < kRedBlocks; ++b) a += part[(int64_t)blockIdx.x * kRedBlocks + b];
    out[blockIdx.x] = a;
}

// T1 = G * RES per sample (score!, multivariate.jl:67); optionally * w and upper-triangular only
__global__ void __launch_bounds__(256)
k_mv_apply(const double *__restrict__ IN, int64_t n, int r, RMat G, int upper, const double *__restrict__ w,
           double *__restrict__ OUT, double *__restrict__ part /* sum of squares per block, or null */)
{
    int64_t s = blockIdx.x * 256ll + threadIdx.x;
    double v[1] = {0.0};
    if (s < n) {
        double in[kMaxR];
        double wt = w ? w[s] : 1.0;
        for (int l = 0; l < r; ++l) in[l] = IN[(int64_t)l * n + s] * wt;
        for (int i = 0; i < r; ++i) {
            double a = 0.0;
            for (int l = upper ? i : 0; l < r; ++l) a += G.v[i + r * l] * in[l];
            if (OUT) OUT[(int64_t)i * n + s] = a;
            v[0] += a * a;
        }
    }
    if (part) block_sum<1>(v, part + blockIdx.x);
}

// the same with the trait count known at compile time: the sample's r values stay in registers (the runtime-r kernel above
// indexes a local array dynamically, i.e. through scratch memory: 70 us instead of 20 us at r = 10, n = 500k)
template <int R>
__global__ void __launch_bounds__(256)
k_mv_apply_t(const double *__restrict__ IN, int64_t n, RMat G, int upper, const double *__restrict__ w,
             double *__restrict__ OUT, double *__restrict__ part)
{
    int64_t s = blockIdx.x * 256ll + threadIdx.x;
    double v[1] = {0.0};
    if (s < n) {
        double in[R];
        double wt = w ? w[s] : 1.0;
        #pragma unroll
        for (int l = 0; l < R; ++l) in[l] = IN[(int64_t)l * n + s] * wt;
        #pragma unroll
        for (int i = 0; i < R; ++i) {
            double a = 0.0;
            #pragma unroll
            for (int l = 0; l < R; ++l) if (l >= (upper ? i : 0)) a += G.v[i + R * l] * in[l];
            if (OUT) OUT[(int64_t)i * n + s] = a;
            v[0] += a * a;
        }
    }
    if (part) block_sum<1>(v, part + blockIdx.x);
}
static void launch_mv_apply(unsigned nb, hipStream_t s, const double *IN, int64_t n, int r, const RMat &G, int upper, const double *w,
                            double *OUT, double *part)
{
#define MIH_APPLY(RV) case RV: hipLaunchKernelGGL((k_mv_apply_t<RV>), dim3(nb), dim3(256), 0, s, IN, n, G, upper, w, OUT, part); return;
    switch (r) {
        MIH_APPLY(1) MIH_APPLY(2) MIH_APPLY(3) MIH_APPLY(4) MIH_APPLY(5) MIH_APPLY(6) MIH_APPLY(7) MIH_APPLY(8)
        MIH_APPLY(9) MIH_APPLY(10) MIH_APPLY(11) MIH_APPLY(12) MIH_APPLY(13) MIH_APPLY(14) MIH_APPLY(15) MIH_APPLY(16)
        default: break;
    }
#undef MIH_APPLY
    hipLaunchKernelGGL(k_mv_apply, dim3(nb), dim3(256), 0, s, IN, n, r, G, upper, w, OUT, part);
}

// vectorize!(full_b, B, C) after the axpy (multivariate.jl:99-113): full[i + r*j] = eta * df[i][j]
// The covariate tail full[r*p .. r*p + rq) rides along as a kernel argument (block 0 writes it) instead of a separate copy.
__global__ void __launch_bounds__(256)
k_mv_full(const double *__restrict__ DF, int64_t p, int r, double eta, double *__restrict__ full, CMat tail, int rq)
{
    extern __shared__ double tile[];                       // [r][257]: 256 SNPs of every trait plane, read and written coalesced
    if (blockIdx.x == 0) for (int t = threadIdx.x; t < rq; t += 256) full[(int64_t)r * p + t] = tail.v[t];
    const int64_t j0 = blockIdx.x * 256ll;
    const int cnt = (int)(p - j0 < 256 ? p - j0 : 256);
    for (int i = 0; i < r; ++i)
        if ((int)threadIdx.x < cnt) tile[i * 257 + threadIdx.x] = eta * DF[(int64_t)i * p + j0 + threadIdx.x];
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * r; t += 256) {
        const int jj = t / r, i = t - jj * r;
        full[(int64_t)r * j0 + t] = tile[i * 257 + jj];
    }
}
__global__ void k_mv_scatter(const int64_t *__restrict__ li, const double *__restrict__ val, int64_t nnz,
                             const double *__restrict__ DF, int64_t p, int r, double eta, double *__restrict__ full)
{
    int64_t t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= nnz) return;
    int64_t l = li[t], j = l / r; int i = (int)(l - j * r);
    full[l] = fma(eta, DF[(int64_t)i * p + j], val[t]);
}
// unvectorize! of the projected gradient at init (multivariate.jl:438-440): df[i][j] = full[i + r*j]
__global__ void k_mv_unvec(const double *__restrict__ full, int64_t p, int r, double *__restrict__ DF)
{
    int64_t t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= p * r) return;
    int64_t j = t / r; int i = (int)(t - j * r);
    DF[(int64_t)i * p + j] = full[t];
}
__global__ void k_mv_gather(const double *__restrict__ DF, int64_t p, int r, const int64_t *__restrict__ cols, int64_t nc,
                            double *__restrict__ out /* [r][nc] */)
{
    int64_t t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= nc * r) return;
    int64_t i = t / nc, c = t - i * nc;
    out[t] = DF[i * p + cols[c]];
}
// column shard: the projected gradient of init_iht_indices! rebuilt from the survivors of the GLOBAL threshold (the local projection
// only knows the shard's own): df[i][j] = value of the entry i + r*j
__global__ void k_mv_put_df(const int64_t *__restrict__ li, const double *__restrict__ val, int64_t nnz, int64_t p, int r, double *__restrict__ DF)
{
    int64_t t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= nnz) return;
    int64_t l = li[t], j = l / r; int i = (int)(l - j * r);
    DF[(int64_t)i * p + j] = val[t];
}
// column shard: this shard's share of |df_S|^2 (the numerator of iht_stepsize!, multivariate.jl:247), left behind the n*r
// partial products so that ONE all-reduce sums both over the shards
__global__ void __launch_bounds__(256) k_mv_sumsq(const double *__restrict__ v, int64_t cnt, double *__restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ double sh[256];
    double a = 0.0;
    for (int64_t t = threadIdx.x; t < cnt; t += 256) a += v[t] * v[t];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) *out = sh[0];
}
__global__ void k_mv_copy1(const double *__restrict__ src, double *__restrict__ dst) { *dst = *src; }
// predict!(v::mIHTVariable) (cross_validation.jl:288-299): sum (Y - mu)^2 * cv_wts
__global__ void __launch_bounds__(256)
k_mv_mse(const double *__restrict__ Y, const double *__restrict__ MU, const double *__restrict__ w, int64_t n, int r,
         double *__restrict__ part)
{
    int64_t s = blockIdx.x * 256ll + threadIdx.x;
    double v[1] = {0.0};
    if (s < n) for (int i = 0; i < r; ++i) { double d = Y[(int64_t)i * n + s] - MU[(int64_t)i * n + s]; v[0] += d * d * w[s]; }
    block_sum<1>(v, part + blockIdx.x);
}

// ---- small dense r x r algebra on the host (column-major) -----------------------------------
static bool chol_upper(std::vector<double> &A, int r)
{
    for (int j = 0; j < r; ++j) {
        double d = A[j + r * j];
        for (int l = 0; l < j; ++l) d -= A[l + r * j] * A[l + r * j];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d); A[j + r * j] = d;
        for (int i = j + 1; i < r; ++i) {
            double s = A[j + r * i];
            for (int l = 0; l < j; ++l) s -= A[l + r * j] * A[l + r * i];
            A[j + r * i] = s / d;
        }
    }
    return true;
}
// inv!(cholesky!(Symmetric(A, :U))) (multivariate.jl:280): full symmetric inverse
static bool spd_inverse(std::vector<double> &A, int r)
{
    if (!chol_upper(A, r)) return false;
    std::vector<double> Ui((size_t)r * r, 0.0);
    for (int j = 0; j < r; ++j) {
        Ui[j + r * j] = 1.0 / A[j + r * j];
        for (int i = j - 1; i >= 0; --i) {
            double s = 0.0;
            for (int l = i + 1; l <= j; ++l) s += A[i + r * l] * Ui[l + r * j];
            Ui[i + r * j] = -s / A[i + r * i];
        }
    }
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < r; ++j) {
            double s = 0.0;
            for (int l = std::max(i, j); l < r; ++l) s += Ui[i + r * l] * Ui[j + r * l];
            A[i + r * j] = s;
        }
    return true;
}
// cholesky!(Symmetric(G,:U), Val(true)); triu!(G) (multivariate.jl:241-242): LAPACK dpstf2 with
// tol = 0 (diagonal pivoting, first maximum); the permutation is dropped as the reference does.
static void pivoted_chol_triu(std::vector<double> &G, int r)
{
    std::vector<double> dot(r, 0.0);
    for (int j = 0; j < r; ++j) {
        int pvt = j; double ajj = -std::numeric_limits<double>::infinity();
        for (int i = j; i < r; ++i) {
            if (j > 0) dot[i] += G[(j - 1) + r * i] * G[(j - 1) + r * i];
            double d = G[i + r * i] - dot[i];
            if (d > ajj) { ajj = d; pvt = i; }
        }
        if (!(ajj > 0.0)) { G[j + r * j] = ajj; break; }
        if (pvt != j) {
            G[pvt + r * pvt] = G[j + r * j];
            for (int l = 0; l < j; ++l) std::swap(G[l + r * j], G[l + r * pvt]);
            for (int i = pvt + 1; i < r; ++i) std::swap(G[j + r * i], G[pvt + r * i]);
            for (int i = j + 1; i < pvt; ++i) std::swap(G[j + r * i], G[i + r * pvt]);
            std::swap(dot[j], dot[pvt]);
        }
        ajj = std::sqrt(ajj); G[j + r * j] = ajj;
        for (int i = j + 1; i < r; ++i) {
            double s = G[j + r * i];
            for (int l = 0; l < j; ++l) s -= G[l + r * j] * G[l + r * i];
            G[j + r * i] = s / ajj;
        }
    }
    for (int j = 0; j < r; ++j) for (int i = j + 1; i < r; ++i) G[i + r * j] = 0.0;
}
// LU with partial pivoting: log|det|, sign, optional inverse
static bool lu_logdet_inverse(const std::vector<double> &Ain, int r, double *logabsdet, int *sign, std::vector<double> *inv)
{
    std::vector<double> A = Ain; std::vector<int> piv(r);
    int sg = 1; double lad = 0.0;
    for (int j = 0; j < r; ++j) {
        int pv = j; double mx = std::fabs(A[j + r * j]);
        for (int i = j + 1; i < r; ++i) if (std::fabs(A[i + r * j]) > mx) { mx = std::fabs(A[i + r * j]); pv = i; }
        piv[j] = pv;
        if (pv != j) { sg = -sg; for (int c = 0; c < r; ++c) std::swap(A[j + r * c], A[pv + r * c]); }
        double d = A[j + r * j];
        if (d == 0.0) return false;
        if (d < 0) sg = -sg;
        lad += std::log(std::fabs(d));
        for (int i = j + 1; i < r; ++i) A[i + r * j] /= d;
        for (int c = j + 1; c < r; ++c) { double f = A[j + r * c]; for (int i = j + 1; i < r; ++i) A[i + r * c] -= A[i + r * j] * f; }
    }
    if (inv) {
        inv->assign((size_t)r * r, 0.0);
        for (int c = 0; c < r; ++c) {
            double *col = inv->data() + (size_t)r * c;
            for (int i = 0; i < r; ++i) col[i] = (i == c) ? 1.0 : 0.0;
            for (int j = 0; j < r; ++j) if (piv[j] != j) std::swap(col[j], col[piv[j]]);
            for (int j = 0; j < r; ++j) for (int i = j + 1; i < r; ++i) col[i] -= A[i + r * j] * col[j];
            for (int j = r - 1; j >= 0; --j) { col[j] /= A[j + r * j]; for (int i = 0; i < j; ++i) col[i] -= A[i + r * j] * col[j]; }
        }
    }
    if (logabsdet) *logabsdet = lad;
    if (sign) *sign = sg;
    return true;
}

// One mIHTVariable (src/data_structures.jl:140-180), device-resident.
struct MvVar {
    Arena arena;                                         // first member: outlives the buffers carved out of it (common.h)
    const mih_mat *h = nullptr;
    int64_t n = 0, p = 0, k = 0; int q = 0, r = 0;
    std::vector<uint8_t> zkeep; int64_t zkeepn = 0;      // r * sum(zkeep)
    const double *Y_host = nullptr, *Z_host = nullptr;
    int init_beta = 0;
    XtvTune tune;
    hipStream_t s = nullptr;
    DevBuf<double> Y, Z, w, BX, MU, RES, T1, DF, full, red, scal, gval, tmpn;
    DevBuf<int64_t> sidx; DevBuf<double> sval; DevBuf<uint8_t> mask; DevBuf<int32_t> pairs;
    // small results come home through a polled flag (SpinFlag, common.h); the column list and the [trait][column] coefficient
    // matrix go up from pinned memory (every upload is followed by a readback before the next one overwrites the buffer)
    SpinFlag flag; PinBuf<double> hpin; PinBuf<double> upin;
    int readback(const double *src_dev, size_t count, double *dst)
    {
        if (count > hpin.n) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(hpin.alloc(count * 2 + 64, true)); }
        MIH_TRY(readback_words(s, flag, reinterpret_cast<const uint64_t *>(src_dev), reinterpret_cast<uint64_t *>(hpin.p), count));
        stage.synced();                                     // everything queued before has run: the staging ring's slots are free again
        std::memcpy(dst, hpin.p, sizeof(double) * count);
        return MIH_OK;
    }
    // Two small host arrays reach the device through a slot of a pinned ring and ONE small kernel (HostStage, common.h) instead of two
    // copy operations; lists that do not fit a slot go the old way.
    HostStage stage;
    int upload_pair(const void *a, size_t bytes_a, void *dst_a, const void *b, size_t bytes_b, void *dst_b)
    {
        const uint64_t *pin = nullptr;
        MIH_TRY(stage.put(s, a, bytes_a, b, bytes_b, &pin));
        if (pin) { stage_to_device(s, pin, reinterpret_cast<uint64_t *>(dst_a), (bytes_a + 7) / 8, reinterpret_cast<uint64_t *>(dst_b), (bytes_b + 7) / 8); return MIH_OK; }
        const size_t words = (bytes_a + 7) / 8 + (bytes_b + 7) / 8;
        if (words > upin.n) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(upin.alloc(words * 2 + 64)); }
        else MIH_HIP(hipStreamSynchronize(s));              // (rare path: the one pinned buffer may still be in flight)
        if (bytes_a) std::memcpy(upin.p, a, bytes_a);
        if (bytes_b) std::memcpy(upin.p + (bytes_a + 7) / 8, b, bytes_b);
        if (bytes_a) MIH_HIP(hipMemcpyAsync(dst_a, upin.p, bytes_a, hipMemcpyHostToDevice, s));
        if (bytes_b) MIH_HIP(hipMemcpyAsync(dst_b, upin.p + (bytes_a + 7) / 8, bytes_b, hipMemcpyHostToDevice, s));
        return MIH_OK;
    }
    int upload_cols_coef(const std::vector<double> &coef)      // sidx <- cols, mcoef <- coef
    {
        return upload_pair(cols.data(), sizeof(int64_t) * cols.size(), sidx.p, coef.data(), sizeof(double) * coef.size(), mcoef.p);
    }
    // The step size of the NEXT step is computed at the end of a step (step_tail): everything iht_stepsize! needs -- the support,
    // df on it, Gamma -- is final then.  What came home: the denominator, for which support, with which factor of Gamma.
    bool spec_ok = false; double spec_denom = 0.0; std::vector<int64_t> spec_cols; std::vector<double> spec_U;
    DevBuf<int32_t> pairs_s; bool pairs_ok = false, pairs_s_ok = false;      // the index-pair tables (Gram: a <= b; score: (trait, covariate)) go up once
    XtvWork xtv; XvWork xv; TopkWork topk;
    int nb = 0;
    // host
    Sparse B, B0, best_B;                  // linear index i + r*j
    std::vector<int64_t> cols;             // idx: columns with a non-zero (sorted)
    std::vector<double> dfcols;            // df[:, idx] as [r][ncols]
    DevBuf<double> mcoef; std::vector<double> xbcoef;      // [r][ncols] coefficient matrices of the multi-trait X*v
    std::vector<double> C, C0, best_C, df2, G, G0, gram;
    std::vector<uint8_t> idc;
    int64_t nsamples = 0;
    bool choose_fired = false;
    int (*choose_cb)(void *, int32_t, const int64_t *, int64_t, int64_t, int64_t *) = nullptr;   // mih_fit_params::choose
    void *choose_user = nullptr;

    bool own_stream = true;
    ~MvVar() { if (s) (void)hipStreamSynchronize(s); if (s && own_stream) (void)hipStreamDestroy(s); }       // see ~IhtVar

    // Column-sharded fit (mih_comm, as the univariate IhtVar of fit_state.h): this process owns columns [col0, col0 + p) of pg; Y, Z
    // and every n x r matrix are replicated, so the residuals, the Gram matrix, Gamma, the loglikelihood and T1 = Gamma * resid are
    // computed redundantly and identically on every rank and the X'R pass needs no exchange.  Per iteration: one all-reduce of
    // n r + 1 doubles (X_S df_S of iht_stepsize!, the shards' |df_S|^2 riding as the last element), one of n r doubles per
    // update_xb!, and one all-gather of 1 + 2K doubles per projection -- the shards' top-K entries of vec(B) as (global linear
    // index, value) pairs, after which every rank holds the WHOLE k-sparse model of the step (Bg): _choose!'s count and rule and
    // check_convergence's maxima are then computed locally and identically everywhere.
    const mih_comm *comm = nullptr;
    int64_t col0 = 0, pg = 0;
    Sparse Bg, B0g;                        // the whole model, global linear index i + r*(col0 + j); identical on every rank
    double spec_numer = 0.0;               // |df_S|^2 over all shards, home with the step-size denominator
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    int comm_fail(int rc) { set_error("communicator callback failed (%d)", rc); return MIH_BAD_ARG; }
    // sum over the shards of a device vector: queued on this stream when the communicator is the library's own, else through
    // the caller's callback (which is handed the device pointer) after a stream synchronisation.  Timed by kind as in fit_state.h
    // (0: with the step size's numerator riding along, 1: plain products)
    int allreduce_dev(double *buf, int64_t cnt, int kind)
    {
        Profile &pf = *h->prof;
        ExchRecord rec; rec.kind = kind;
        const bool timed = pf.on && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess;
        if (timed) (void)hipEventRecord(rec.e0, s);
        const int nrc = comm_native_allreduce_on_stream(comm, buf, cnt, 0, s, h->device);
        if (nrc >= 0) {
            if (timed) { (void)hipEventRecord(rec.e1, s); std::lock_guard<std::mutex> g(pf.mu); pf.xopen.push_back(rec); }
            return nrc;
        }
        if (timed) { (void)hipEventDestroy(rec.e0); (void)hipEventDestroy(rec.e1); }
        const double t0 = now_ms();
        MIH_HIP(hipStreamSynchronize(s));
        int rc = comm->allreduce(comm->user, buf, cnt, 0, 1);
        pf.exch_host(kind, now_ms() - t0);
        return rc ? comm_fail(rc) : MIH_OK;
    }
    int allgather_host(const double *send, int64_t cnt, std::vector<double> &recv)
    {
        recv.assign((size_t)cnt * comm->world, 0.0);
        const double t0 = now_ms();
        int rc = comm->allgather(comm->user, send, cnt, recv.data());
        h->prof->exch_host(2, now_ms() - t0);
        return rc ? comm_fail(rc) : MIH_OK;
    }

    // shared_stream != null: one of a lock-step batch (mih_cv_mv): it runs on the batch's stream and leaves the
    // X'R pass to the batch driver
    int create(const mih_mat *hh, const mih_fit_params *prm, const double *Yh, int64_t rr, const double *Zh, int64_t qq,
               hipStream_t shared_stream = nullptr)
    {
        h = hh; n = h->n; p = h->p; r = (int)rr; q = (int)qq; k = prm->k; Y_host = Yh; Z_host = Zh; init_beta = prm->init_beta; tune = xtv_tune(prm);
        choose_cb = prm->comm ? nullptr : prm->choose; choose_user = prm->choose_user;
        comm = prm->comm; pg = p; col0 = 0;
        if (comm) {
            if (!comm->allreduce || !comm->allgather || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world ||
                comm->col_offset < 0 || comm->col_offset + p > comm->p_global) {
                set_error("invalid mih_comm (callbacks, rank/world or column range)"); return MIH_BAD_ARG;
            }
            if (shared_stream) { set_error("cross-validation shards over (fold,k) combinations (rank/world), not over columns"); return MIH_BAD_ARG; }
            if (init_beta) { set_error("init_beta is not available for a column-sharded multivariate fit"); return MIH_BAD_ARG; }
            col0 = comm->col_offset; pg = comm->p_global;
        }
        if (r < 1 || r > kMaxR) { set_error("number of traits r=%d must be in 1..%d", r, kMaxR); return MIH_BAD_DIM; }
        if (q < 1 || r * q > kMaxRQ) { set_error("r*q = %d exceeds %d", r * q, kMaxRQ); return MIH_BAD_DIM; }
        zkeep.resize(q); int64_t zs = 0;
        for (int l = 0; l < q; ++l) { zkeep[l] = prm->zkeep ? (prm->zkeep[l] != 0) : 1; zs += zkeep[l]; }
        zkeepn = (int64_t)r * zs;
        if (shared_stream) { s = shared_stream; own_stream = false; }
        else MIH_HIP(hipStreamCreate(&s));
        nb = (int)nblk(n);
        size_t rn = (size_t)r * n;
        int64_t kcap = std::max<int64_t>(k + (int64_t)r * q, 64) + 1024;
        {
            const size_t redn = std::max<size_t>((size_t)nb, (size_t)kMaxR * kMaxR * kRedBlocks);
            size_t dev = sizeof(double) * (5 * rn + 8 + (size_t)q * n + 2 * (size_t)n + (size_t)r * p + (size_t)r * (p + q) + redn + (size_t)kMaxR * kMaxR + 64
                                           + 2 * (size_t)kcap + (size_t)kcap * r) + (size_t)n + sizeof(int32_t) * 2 * (size_t)kMaxR * kMaxR
                         + xv_work_bytes(h, kcap, kcap - 1024) + sizeof(uint32_t) * 2048 + 64 + 16 * ((size_t)kcap + 1025) + 16 * (4096 + 1) + 256 + 40 * 256;
            size_t pin = sizeof(uint64_t) * (2 + 2 * ((size_t)kcap + 64) + 16 + HostStage::kSlots * ((size_t)kcap * (r + 1) + 8)) + 6 * 256;
            MIH_TRY(arena.reserve(dev, pin));
        }
        ArenaScope in_arena(&arena);
        MIH_TRY(Y.alloc(rn)); MIH_TRY(Z.alloc((size_t)q * n)); MIH_TRY(w.alloc(n)); MIH_TRY(BX.alloc(rn)); MIH_TRY(MU.alloc(rn));
        MIH_TRY(RES.alloc(rn)); MIH_TRY(T1.alloc(rn + 8) /* + the step size's numerator behind a shard's partial products */); MIH_TRY(DF.alloc((size_t)r * p)); MIH_TRY(full.alloc((size_t)r * (p + q)));
        MIH_TRY(red.alloc(std::max<size_t>((size_t)nb, (size_t)kMaxR * kMaxR * kRedBlocks))); MIH_TRY(scal.alloc((size_t)kMaxR * kMaxR + 64));
        MIH_TRY(tmpn.alloc(n)); MIH_TRY(mask.alloc(n)); MIH_TRY(pairs.alloc(2 * (size_t)kMaxR * kMaxR));
        MIH_TRY(sidx.alloc(kcap)); MIH_TRY(sval.alloc(kcap)); MIH_TRY(gval.alloc((size_t)kcap * r));
        MIH_TRY(stage.init((size_t)kcap * (r + 1) + 8));
        if (own_stream) { ArenaScope own_buffers(nullptr); MIH_TRY(xtv_work_init(h, xtv, r, tune)); }
        MIH_TRY(xv_work_init(h, xv, kcap, kcap - 1024)); MIH_TRY(topk_work_init(topk, kcap));
        // Y (r x n) and Z (q x n) column-major -> planes
        std::vector<double> pl(std::max(rn, (size_t)q * n));
        for (int64_t sidx_ = 0; sidx_ < n; ++sidx_) for (int i = 0; i < r; ++i) pl[(size_t)i * n + sidx_] = Yh[i + (size_t)r * sidx_];
        MIH_HIP(hipMemcpy(Y.p, pl.data(), sizeof(double) * rn, hipMemcpyHostToDevice));
        for (int64_t sidx_ = 0; sidx_ < n; ++sidx_) for (int l = 0; l < q; ++l) pl[(size_t)l * n + sidx_] = Zh[l + (size_t)q * sidx_];
        MIH_HIP(hipMemcpy(Z.p, pl.data(), sizeof(double) * (size_t)q * n, hipMemcpyHostToDevice));
        C.assign((size_t)r * q, 0.0); C0 = C; best_C = C; df2 = C; idc.assign(q, 0);
        G.assign((size_t)r * r, 0.0); G0 = G; gram = G;
        return MIH_OK;
    }

    int ensure_stage(int64_t nnz)
    {
        if ((size_t)nnz <= sidx.n) return MIH_OK;
        MIH_HIP(hipStreamSynchronize(s));
        MIH_TRY(sidx.alloc((size_t)nnz * 2)); MIH_TRY(sval.alloc((size_t)nnz * 2)); MIH_TRY(gval.alloc((size_t)nnz * 2 * r));
        return MIH_OK;
    }
    CMat cmat(const std::vector<double> &c) const { CMat o; for (int t = 0; t < kMaxRQ; ++t) o.v[t] = t < r * q ? c[t] : 0.0; return o; }
    RMat rmat(const std::vector<double> &g) const { RMat o; for (int t = 0; t < kMaxR * kMaxR; ++t) o.v[t] = t < r * r ? g[t] : 0.0; return o; }

    int set_weights(const uint8_t *m, int invert)
    {
        if (!m) {
            std::vector<double> ones(n, invert ? 0.0 : 1.0);
            MIH_HIP(hipMemcpy(w.p, ones.data(), sizeof(double) * n, hipMemcpyHostToDevice));
            return MIH_OK;
        }
        MIH_HIP(hipMemcpyAsync(mask.p, m, n, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_mask_to_wts, dim3(nblk(n)), dim3(256), 0, s, mask.p, n, invert, w.p);
        return MIH_OK;
    }
    void update_cols()         // update_support!(v.idx, v.B) (multivariate.jl:197-206)
    {
        cols.clear();
        for (size_t t = 0; t < B.idx.size(); ++t) { int64_t j = B.idx[t] / r; if (cols.empty() || cols.back() != j) cols.push_back(j); }
    }
    // BX = B[:,idx] * X[idx,:] (multivariate.jl:21-31), one sparse X*v per trait
    int update_xb()
    {
        update_cols();
        int64_t nc = (int64_t)cols.size();
        MIH_TRY(ensure_stage(nc));
        if (!nc) {
            MIH_HIP(hipMemsetAsync(BX.p, 0, sizeof(double) * (size_t)n * r, s));
            return comm ? allreduce_dev(BX.p, n * r, 1) : MIH_OK;          // (a shard without a support column still joins the sum)
        }
        // coefficient matrix [trait][support column] (zero where a trait does not use the column), one upload, one launch
        xbcoef.assign((size_t)nc * r, 0.0);
        size_t c = 0;
        for (size_t t = 0; t < B.idx.size(); ++t) {
            int64_t j = B.idx[t] / r; int ii = (int)(B.idx[t] - j * r);
            while (cols[c] != j) ++c;
            xbcoef[(size_t)ii * nc + c] = B.val[t];
        }
        if ((size_t)nc * r > mcoef.n) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(mcoef.alloc((size_t)nc * r * 2)); }
        MIH_TRY(upload_cols_coef(xbcoef));
        MIH_TRY(xv_sparse_multi_device(h, xv, sidx.p, mcoef.p, nc, r, BX.p, s, cols.data()));
        return comm ? allreduce_dev(BX.p, n * r, 1) : MIH_OK;              // the shards' partial products
    }
    // update_mu! + update_resid! (+ Gram matrix resid*resid' for solve_Sigma!/loglikelihood)
    int resid_and_gram()
    {
        hipLaunchKernelGGL(k_mv_resid, dim3(nb), dim3(256), 0, s, Y.p, Z.p, BX.p, w.p, n, r, q, cmat(C), MU.p, RES.p);
        std::vector<int32_t> pr;
        for (int a = 0; a < r; ++a) for (int b = a; b < r; ++b) pr.push_back(a);
        size_t np_ = pr.size();
        for (int a = 0; a < r; ++a) for (int b = a; b < r; ++b) pr.push_back(b);
        if (!pairs_ok) { MIH_HIP(hipMemcpyAsync(pairs.p, pr.data(), sizeof(int32_t) * pr.size(), hipMemcpyHostToDevice, s)); pairs_ok = true; }
        hipLaunchKernelGGL(k_mv_dots, dim3(kRedBlocks, (unsigned)np_), dim3(256), 0, s, RES.p, RES.p, n, pairs.p, pairs.p + np_, red.p);
        hipLaunchKernelGGL(k_mv_dots_final, dim3((unsigned)np_), dim3(64), 0, s, red.p, scal.p);
        std::vector<double> g(np_);
        MIH_TRY(readback(scal.p, np_, g.data()));
        MIH_HIP(hipGetLastError());                  // a failed launch anywhere in this iteration's chain surfaces here
        size_t t = 0;
        for (int a = 0; a < r; ++a) for (int b = a; b < r; ++b, ++t) { gram[a + r * b] = g[t]; gram[b + r * a] = g[t]; }
        return MIH_OK;
    }
    // solve_Sigma! (multivariate.jl:276-282): Gamma = inv(resid*resid' / nsamples)
    int solve_sigma()
    {
        MIH_TRY(resid_and_gram());
        std::vector<double> a = gram;
        for (auto &x : a) x /= (double)nsamples;
        if (!spd_inverse(a, r)) { set_error("residual covariance is not positive definite"); return MIH_NAN_LOGL; }
        G = a;
        return MIH_OK;
    }
    // loglikelihood (multivariate.jl:9-13): n/2 logdet(Gamma) - 1/2 tr(Gamma * resid*resid')
    double loglik() const
    {
        double tr = 0.0;
        for (int a = 0; a < r; ++a) { double sacc = 0.0; for (int l = 0; l < r; ++l) sacc += G[a + r * l] * gram[l + r * a]; tr += sacc; }
        double lad; int sg;
        if (!lu_logdet_inverse(G, r, &lad, &sg, nullptr) || sg < 0) return std::numeric_limits<double>::quiet_NaN();
        return (double)nsamples / 2.0 * lad - 0.5 * tr;
    }
    // score! (multivariate.jl:66-92) = score_pre (T1 = Gamma * resid) ; DF = X' T1' ; score_post (df2 = T1 Z')
    int score()
    {
        MIH_TRY(score_pre());
        MIH_TRY(pass());
        return score_post();
    }
    // The r-trait X'R pass.  xtv_digits = -1 (round 5, opt-in as in the lock-step drivers): when EVERY trait's row of T1 passes the
    // guard max |t| <= 128 rms(t) the pass takes the 43-bit fixed-point format (four residuals per operand: three operands instead
    // of four at r = 10), else the 54-bit default -- decided per pass from this fit's own T1, at the price of one more readback.
    DevBuf<double> guard;
    int pass()
    {
        if (own_stream && tune.digits == -1 && xtv.has_alt) {
            if (guard.n < (size_t)(2 * nb + 1) * r) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(guard.alloc((size_t)(2 * nb + 1) * r)); }
            double *flags = guard.p + (size_t)2 * nb * r;
            for (int i = 0; i < r; ++i) {
                hipLaunchKernelGGL(k_r_guard, dim3(nb), dim3(256), 0, s, T1.p + (size_t)i * n, n, guard.p + (size_t)2 * nb * i);
                hipLaunchKernelGGL(k_r_guard_final, dim3(1), dim3(256), 0, s, guard.p + (size_t)2 * nb * i, nb, n, flags + i);
            }
            std::vector<double> f((size_t)r);
            MIH_TRY(readback(flags, (size_t)r, f.data()));
            bool all = true;
            for (int i = 0; i < r; ++i) all = all && f[(size_t)i] == 1.0;
            xtv.use_alt = all;
            if (all) h->prof->count(MIH_CNT_RESIDUALS_43BIT, r);
        }
        const int rc = xtv_device(h, xtv, T1.p, r, DF.p, s);
        xtv.use_alt = false;
        return rc;
    }
    int score_pre()
    {
        launch_mv_apply((unsigned)nb, s, RES.p, n, r, rmat(G), 0, (const double *)nullptr, T1.p, (double *)nullptr);
        return MIH_OK;
    }
    int score_post()
    {
        std::vector<int32_t> pr;
        for (int l = 0; l < q; ++l) for (int i = 0; i < r; ++i) pr.push_back(i);
        size_t np_ = pr.size();
        for (int l = 0; l < q; ++l) for (int i = 0; i < r; ++i) pr.push_back(l);
        if (!pairs_s_ok) {
            MIH_TRY(pairs_s.alloc(pr.size()));
            MIH_HIP(hipMemcpyAsync(pairs_s.p, pr.data(), sizeof(int32_t) * pr.size(), hipMemcpyHostToDevice, s));
            pairs_s_ok = true;
        }
        hipLaunchKernelGGL(k_mv_dots, dim3(kRedBlocks, (unsigned)np_), dim3(256), 0, s, T1.p, Z.p, n, pairs_s.p, pairs_s.p + np_, red.p);
        hipLaunchKernelGGL(k_mv_dots_final, dim3((unsigned)np_), dim3(64), 0, s, red.p, scal.p);
        return readback(scal.p, np_, df2.data());   // [i + r*l]
    }
    // The end of a step (or of the initialisation) and the beginning of the next step in ONE host synchronisation (the univariate
    // step_post_fused, fit_state.h): [with_df2: df2 = T1 Z' (score!, multivariate.jl:88-91)], df on the support (gathered on the
    // device, [trait][column]) and the whole iht_stepsize! of the next step -- X_S df_S straight from the device copy of df_S,
    // the pivoted Cholesky factor of Gamma, the weighted sum of squares -- are queued back to back and come home in one copy:
    // [df_S | df2 | sum].  Two synchronisations and three small copies less per iteration than three separate round trips; the
    // step-size product is wasted only when the fit stops here.  T1 (= Gamma * resid, the pass's input) is read by the df2 dots
    // before the product overwrites it: same stream.
    int step_tail(bool with_df2)
    {
        const int64_t nc = (int64_t)cols.size();
        const size_t rq = (size_t)r * q, ncr = (size_t)nc * r;
        MIH_TRY(ensure_stage(nc));
        if (ncr + rq + 3 > gval.n) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(gval.alloc((ncr + rq + 3) * 2)); }
        double *d_df2 = gval.p + ncr, *d_sum = gval.p + ncr + rq;
        if (with_df2) {
            std::vector<int32_t> pr;
            for (int l = 0; l < q; ++l) for (int i = 0; i < r; ++i) pr.push_back(i);
            size_t np_ = pr.size();
            for (int l = 0; l < q; ++l) for (int i = 0; i < r; ++i) pr.push_back(l);
            if (!pairs_s_ok) {
                MIH_TRY(pairs_s.alloc(pr.size()));
                MIH_HIP(hipMemcpyAsync(pairs_s.p, pr.data(), sizeof(int32_t) * pr.size(), hipMemcpyHostToDevice, s));
                MIH_HIP(hipStreamSynchronize(s));           // (once per variable: `pr` is a local)
                pairs_s_ok = true;
            }
            hipLaunchKernelGGL(k_mv_dots, dim3(kRedBlocks, (unsigned)np_), dim3(256), 0, s, T1.p, Z.p, n, pairs_s.p, pairs_s.p + np_, red.p);
            hipLaunchKernelGGL(k_mv_dots_final, dim3((unsigned)np_), dim3(64), 0, s, red.p, d_df2);
        }
        spec_U = G;
        pivoted_chol_triu(spec_U, r);
        if (nc) {
            MIH_TRY(upload_pair(cols.data(), sizeof(int64_t) * (size_t)nc, sidx.p, nullptr, 0, nullptr));
            hipLaunchKernelGGL(k_mv_gather, dim3(nblk(nc * r)), dim3(256), 0, s, DF.p, p, r, sidx.p, nc, gval.p);
            MIH_TRY(xv_sparse_multi_device(h, xv, sidx.p, gval.p, nc, r, T1.p, s, cols.data()));
        } else MIH_HIP(hipMemsetAsync(T1.p, 0, sizeof(double) * (size_t)n * r, s));
        if (comm) {
            hipLaunchKernelGGL(k_mv_sumsq, dim3(1), dim3(256), 0, s, gval.p, (int64_t)ncr, T1.p + (size_t)n * r);
            MIH_TRY(allreduce_dev(T1.p, n * r + 1, 0));
            hipLaunchKernelGGL(k_mv_copy1, dim3(1), dim3(1), 0, s, T1.p + (size_t)n * r, d_sum + 1);
        }
        launch_mv_apply((unsigned)nb, s, T1.p, n, r, rmat(spec_U), 1, w.p, (double *)nullptr, red.p);
        hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, s, red.p, nb, 1, d_sum);
        std::vector<double> home(ncr + rq + 2);
        MIH_TRY(readback(gval.p, ncr + rq + (comm ? 2 : 1), home.data()));
        if (comm) spec_numer = home[ncr + rq + 1];
        dfcols.assign(home.begin(), home.begin() + (std::ptrdiff_t)ncr);
        if (with_df2) for (size_t t = 0; t < rq; ++t) df2[t] = home[ncr + t];
        spec_denom = home[ncr + rq]; spec_cols = cols; spec_ok = true;
        return MIH_OK;
    }
    int gather_df_cols()
    {
        int64_t nc = (int64_t)cols.size();
        dfcols.assign((size_t)nc * r, 0.0);
        if (!nc) return MIH_OK;
        MIH_TRY(ensure_stage(nc));
        MIH_HIP(hipMemcpyAsync(sidx.p, cols.data(), sizeof(int64_t) * nc, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_mv_gather, dim3(nblk(nc * r)), dim3(256), 0, s, DF.p, p, r, sidx.p, nc, gval.p);
        return readback(gval.p, (size_t)nc * r, dfcols.data());
    }
    // iht_stepsize! (multivariate.jl:220-254): covariates ignored, pivoted Cholesky of Gamma
    int stepsize(double *eta)
    {
        int64_t nc = (int64_t)cols.size();
        double numer = 0.0;
        for (double x : dfcols) numer += x * x;
        // (column shard: every rank must take the same branch -- the collectives below are issued by all or none -- so the decision
        // rests on spec_ok alone, which every rank sets and clears at the same places; the comparison of the LOCAL column lists
        // holds whenever it does)
        if (spec_ok && (comm || spec_cols == cols)) {                 // computed behind the X'R pass of the previous step (step_tail)
            spec_ok = false;
            G = spec_U;                                               // Gamma is left holding U, as below
            if (comm) numer = spec_numer;
            double e = numer / spec_denom;
            if (std::isinf(e) || std::isnan(e)) e = 1e-8;
            *eta = e;
            return MIH_OK;
        }
        spec_ok = false;
        if (nc) {
            if ((size_t)nc * r > mcoef.n) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(mcoef.alloc((size_t)nc * r * 2)); }
            MIH_TRY(upload_cols_coef(dfcols));                       // dfcols is [trait][column]
            MIH_TRY(xv_sparse_multi_device(h, xv, sidx.p, mcoef.p, nc, r, T1.p, s, cols.data()));
        } else MIH_HIP(hipMemsetAsync(T1.p, 0, sizeof(double) * (size_t)n * r, s));
        if (comm) {
            hipLaunchKernelGGL(k_mv_sumsq, dim3(1), dim3(256), 0, s, mcoef.p, nc * r, T1.p + (size_t)n * r);
            MIH_TRY(allreduce_dev(T1.p, n * r + 1, 0));
            hipLaunchKernelGGL(k_mv_copy1, dim3(1), dim3(1), 0, s, T1.p + (size_t)n * r, scal.p + 1);
        }
        pivoted_chol_triu(G, r);                                   // Gamma is left holding U (fit.jl:230-232 recomputes it)
        launch_mv_apply((unsigned)nb, s, T1.p, n, r, rmat(G), 1, w.p, (double *)nullptr, red.p);
        hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, s, red.p, nb, 1, scal.p);
        double dn[2] = {0.0, 0.0};
        MIH_TRY(readback(scal.p, comm ? 2 : 1, dn));
        const double denom = dn[0];
        if (comm) numer = dn[1];
        double e = numer / denom;
        if (std::isinf(e) || std::isnan(e)) e = 1e-8;
        *eta = e;
        return MIH_OK;
    }
    // the caller shuffles, as the reference does (multivariate.jl:336-348): both lists, then the first `excess` entries go --
    // out of B while it has non-zeros left, then C_nz_idx[i] with the SAME running i
    int choose_by_caller(int64_t excess)
    {
        const int64_t nB = (int64_t)B.idx.size();
        std::vector<int64_t> clist;
        for (int l = 0; l < q; ++l) if (!zkeep[l]) for (int i = 0; i < r; ++i) if (C[i + r * l] != 0.0) clist.push_back(i + (int64_t)r * l);
        const int64_t nC = (int64_t)clist.size();
        std::vector<int64_t> bs((size_t)nB, -1), cs((size_t)nC, -1);
        //  an empty list is not handed over: its data pointer may be NULL, and shuffle! of an empty vector draws nothing
        // from the RNG (its loop runs over 2:length), so the caller's RNG state is what the reference's would be
        if ((nB > 0 && choose_cb(choose_user, MIH_CHOOSE_SHUFFLE_B, B.idx.data(), nB, excess, bs.data()) != 0) ||
            (nC > 0 && choose_cb(choose_user, MIH_CHOOSE_SHUFFLE_C, clist.data(), nC, excess, cs.data()) != 0)) { set_error("the choose callback failed"); return MIH_BAD_ARG; }
        auto is_perm = [](std::vector<int64_t> a, std::vector<int64_t> b) { std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); return a == b; };
        if (!is_perm(bs, B.idx) || !is_perm(cs, clist)) { set_error("the choose callback must return its list in shuffled order"); return MIH_BAD_ARG; }
        std::vector<char> drop((size_t)nB, 0);
        int64_t b_left = nB;
        for (int64_t i = 0; i < excess; ++i) {
            if (b_left > 0) {
                drop[(size_t)(std::lower_bound(B.idx.begin(), B.idx.end(), bs[(size_t)i]) - B.idx.begin())] = 1;
                --b_left;
            } else {
                if (i >= nC) { set_error("_choose!: BoundsError, C_nz_idx[%lld] of %lld (multivariate.jl:344)", (long long)(i + 1), (long long)nC); return MIH_BAD_ARG; }
                C[(size_t)cs[(size_t)i]] = 0.0;
            }
        }
        Sparse kept;
        for (int64_t i = 0; i < nB; ++i) if (!drop[(size_t)i]) { kept.idx.push_back(B.idx[(size_t)i]); kept.val.push_back(B.val[(size_t)i]); }
        B = kept;
        return MIH_OK;
    }
    // _choose!(v::mIHTVariable) (multivariate.jl:310-351): RNG in the reference; without a callback deterministic + flag here
    int choose()
    {
        int64_t cnz = 0;
        for (int l = 0; l < q; ++l) if (!zkeep[l]) for (int i = 0; i < r; ++i) cnz += (C[i + r * l] != 0.0);
        int64_t excess = (int64_t)(comm ? Bg.idx.size() : B.idx.size()) + cnz - (k + zkeepn);
        if (excess <= 0) return MIH_OK;
        choose_fired = true;
        if (choose_cb) return choose_by_caller(excess);
        if (comm) {          // the same deterministic rule over the WHOLE model, which every rank holds; then this shard's part of what is left
            std::vector<size_t> og(Bg.idx.size());
            for (size_t i = 0; i < og.size(); ++i) og[i] = i;
            std::sort(og.begin(), og.end(), [&](size_t a, size_t b) {
                double fa = std::fabs(Bg.val[a]), fb = std::fabs(Bg.val[b]);
                if (fa != fb) return fa < fb;
                return Bg.idx[a] > Bg.idx[b];
            });
            std::vector<char> dg(Bg.idx.size(), 0);
            int64_t t = 0;
            for (; t < excess && t < (int64_t)og.size(); ++t) dg[og[(size_t)t]] = 1;
            Sparse kept;
            for (size_t i = 0; i < Bg.idx.size(); ++i) if (!dg[i]) { kept.idx.push_back(Bg.idx[i]); kept.val.push_back(Bg.val[i]); }
            Bg = kept;
            local_from_global();
            for (int l = 0; l < q && t < excess; ++l) if (!zkeep[l]) for (int i = 0; i < r && t < excess; ++i) if (C[i + r * l] != 0.0) { C[i + r * l] = 0.0; ++t; }
            return MIH_OK;
        }
        std::vector<size_t> ord(B.idx.size());
        for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
            double fa = std::fabs(B.val[a]), fb = std::fabs(B.val[b]);
            if (fa != fb) return fa < fb;
            return B.idx[a] > B.idx[b];
        });
        std::vector<char> drop(B.idx.size(), 0);
        int64_t t = 0;
        for (; t < excess && t < (int64_t)ord.size(); ++t) drop[ord[t]] = 1;
        Sparse nb2;
        for (size_t i = 0; i < B.idx.size(); ++i) if (!drop[i]) { nb2.idx.push_back(B.idx[i]); nb2.val.push_back(B.val[i]); }
        B = nb2;
        for (int l = 0; l < q && t < excess; ++l) if (!zkeep[l]) for (int i = 0; i < r && t < excess; ++i) if (C[i + r * l] != 0.0) { C[i + r * l] = 0.0; ++t; }
        return MIH_OK;
    }
    // project the r(p+q) buffer and split survivors; tail = covariate part of the vector
    int project_full(Sparse &snp, std::vector<double> &ctail, std::vector<uint8_t> &cnz, bool zero_in_place = true)
    {
        if (comm) return project_full_sharded(snp, ctail, cnz);
        std::vector<int64_t> si; std::vector<double> sv;
        MIH_TRY(topk_project_device(full.p, (int64_t)r * (p + q), k + zkeepn, topk, s, si, sv, zero_in_place));
        snp.clear(); cnz.assign((size_t)r * q, 0); ctail.assign((size_t)r * q, 0.0);
        for (size_t t = 0; t < si.size(); ++t) {
            if (si[t] < (int64_t)r * p) { snp.idx.push_back(si[t]); snp.val.push_back(sv[t]); }
            else { ctail[si[t] - (int64_t)r * p] = sv[t]; cnz[si[t] - (int64_t)r * p] = 1; }
        }
        return MIH_OK;
    }
    // project_k!(v) over the shards (IhtVar::project_full_sharded of fit_state.h for vec(B)): the K-th largest |entry| of the whole
    // r(pg + q) vector is the K-th largest of the union of every shard's own top-K and the covariate tail (which every rank
    // holds); ties at that value are kept.  A shard sends its candidates as (global linear index, value) pairs, so every rank ends
    // up with the whole model of the step (Bg).  A shard with more than K entries at or above its own threshold (exact ties
    // there) can send only K: if its smallest sent magnitude still reaches the global threshold the whole model may be
    // incomplete -- the fit stops with an error rather than continue with ranks that disagree.  `full` itself is left as it is.
    int project_full_sharded(Sparse &snp, std::vector<double> &ctail, std::vector<uint8_t> &cnz)
    {
        const int64_t K = k + zkeepn, rq = (int64_t)r * q, lenl = (int64_t)r * p;
        if (K <= 0 || K > (int64_t)r * (pg + q)) { set_error("Attempted to project to sparsity level %lld (vector length %lld)", (long long)K, (long long)((int64_t)r * (pg + q))); return MIH_BAD_ARG; }
        std::vector<double> tail((size_t)rq);
        MIH_HIP(hipMemcpyAsync(tail.data(), full.p + lenl, sizeof(double) * (size_t)rq, hipMemcpyDeviceToHost, s));
        std::vector<int64_t> si; std::vector<double> sv;
        const int64_t Kloc = std::min<int64_t>(K, lenl);
        if (Kloc > 0) MIH_TRY(topk_project_device(full.p, lenl, Kloc, topk, s, si, sv, /*zero_in_place=*/false));
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
        for (int64_t t = 0; t < K && t < (int64_t)ord.size(); ++t) { mine[1 + 2 * t] = (double)(col0 * r + si[ord[(size_t)t]]); mine[2 + 2 * t] = sv[ord[(size_t)t]]; }
        MIH_TRY(allgather_host(mine.data(), slot, all));
        std::vector<double> mags;
        for (int32_t rk = 0; rk < comm->world; ++rk)
            for (int64_t t = 0; t < K; ++t) { const double *e = &all[(size_t)rk * slot + 1 + 2 * t]; if (e[0] >= 0.0) mags.push_back(std::fabs(e[1])); }
        for (int64_t l = 0; l < rq; ++l) mags.push_back(std::fabs(tail[(size_t)l]));
        if ((int64_t)mags.size() < K) { set_error("projection to %lld entries of a vector with %zu non-empty candidates", (long long)K, mags.size()); return MIH_BAD_ARG; }
        std::nth_element(mags.begin(), mags.begin() + (K - 1), mags.end(), std::greater<double>());
        const double a = mags[(size_t)K - 1];
        snp.clear(); cnz.assign((size_t)rq, 0); ctail.assign((size_t)rq, 0.0);
        for (size_t t = 0; t < si.size(); ++t)
            if (std::fabs(sv[t]) >= a) { snp.idx.push_back(si[t]); snp.val.push_back(sv[t]); }
        for (int64_t l = 0; l < rq; ++l)
            if (std::fabs(tail[(size_t)l]) >= a) { ctail[(size_t)l] = tail[(size_t)l]; cnz[(size_t)l] = 1; }
        Bg.clear();
        std::vector<std::pair<int64_t, double>> glob;
        for (int32_t rk = 0; rk < comm->world; ++rk) {
            const double *msg = &all[(size_t)rk * slot];
            double smallest = std::numeric_limits<double>::infinity();
            for (int64_t t = 0; t < K; ++t) {
                if (msg[1 + 2 * t] < 0.0) continue;
                const double mg = std::fabs(msg[2 + 2 * t]);
                smallest = std::min(smallest, mg);
                if (mg >= a) glob.emplace_back((int64_t)msg[1 + 2 * t], msg[2 + 2 * t]);
            }
            if (msg[0] > (double)K && smallest >= a) {               // (decided from the gathered messages: the same on every rank)
                set_error("column-sharded multivariate fit: shard %d holds more than %lld entries tied at the projection's threshold", (int)rk, (long long)K);
                return MIH_BAD_ARG;
            }
        }
        std::sort(glob.begin(), glob.end());
        for (auto &e : glob) { Bg.idx.push_back(e.first); Bg.val.push_back(e.second); }
        return MIH_OK;
    }
    // the shard's part of the whole model: entries of columns [col0, col0 + p), local linear index
    void local_from_global()
    {
        B.clear();
        const int64_t lo = col0 * r, hi = (col0 + p) * r;
        for (size_t t = 0; t < Bg.idx.size(); ++t)
            if (Bg.idx[t] >= lo && Bg.idx[t] < hi) { B.idx.push_back(Bg.idx[t] - lo); B.val.push_back(Bg.val[t]); }
    }
    // _iht_gradstep! + project_k!(v) (multivariate.jl:99-127) from base (Bb, Cb)
    int gradstep(const Sparse &Bb, const std::vector<double> &Cb, double eta)
    {
        std::vector<double> cn((size_t)r * q), tail((size_t)r * q);
        for (int l = 0; l < q; ++l) for (int i = 0; i < r; ++i) {
            cn[i + r * l] = std::fma(eta, df2[i + r * l], Cb[i + r * l]);
            tail[i + r * l] = zkeep[l] ? std::numeric_limits<double>::infinity() : cn[i + r * l];
        }
        hipLaunchKernelGGL(k_mv_full, dim3(nblk(p)), dim3(256), sizeof(double) * 257 * (size_t)r, s, DF.p, p, r, eta, full.p, cmat(tail), r * q);
        MIH_TRY(ensure_stage((int64_t)Bb.idx.size()));
        if (!Bb.idx.empty()) {
            MIH_TRY(upload_pair(Bb.idx.data(), sizeof(int64_t) * Bb.idx.size(), sidx.p, Bb.val.data(), sizeof(double) * Bb.val.size(), sval.p));
            hipLaunchKernelGGL(k_mv_scatter, dim3(nblk((int64_t)Bb.idx.size())), dim3(256), 0, s, sidx.p, sval.p, (int64_t)Bb.idx.size(), DF.p, p, r, eta, full.p);
        }
        Sparse snp; std::vector<double> ct; std::vector<uint8_t> cnz;
        MIH_TRY(project_full(snp, ct, cnz, /*zero_in_place=*/false));       // only the survivor lists are used
        B = snp;
        for (int l = 0; l < q; ++l) for (int i = 0; i < r; ++i)
            C[i + r * l] = zkeep[l] ? cn[i + r * l] : (cnz[i + r * l] ? ct[i + r * l] : 0.0);
        MIH_TRY(choose());
        update_cols();
        for (int l = 0; l < q; ++l) { idc[l] = 0; for (int i = 0; i < r; ++i) if (C[i + r * l] != 0.0) idc[l] = 1; }
        return MIH_OK;
    }
    // initialize_beta!(v::mIHTVariable) + project_k!(v) + update_xb!(v) (multivariate.jl:426-430, 519-558)
    int init_beta_phase(const uint8_t *train)
    {
        DevBuf<double> betad;
        MIH_TRY(betad.alloc((size_t)r * p));
        const double N = (double)nsamples;
        std::vector<double> Sy(r, 0.0), c0sum(r, 0.0);
        for (int64_t j = 0; j < n; ++j) if (!train || train[j]) for (int i = 0; i < r; ++i) Sy[i] += Y_host[i + (size_t)r * j];
        MIH_TRY(init_beta_regress_device(h, w.p, Y.p, r, N, Sy.data(), betad.p, c0sum.data(), red, scal, s, tune));
        for (int l = 1; l < q; ++l) {             // non-genetic covariates 2..q on the host (:547-553)
            double sx = 0.0, sxx = 0.0;
            std::vector<double> sxy(r, 0.0);
            for (int64_t j = 0; j < n; ++j) if (!train || train[j]) {
                double xv = Z_host[l + (size_t)q * j];
                sx += xv; sxx += xv * xv;
                for (int i = 0; i < r; ++i) sxy[i] += xv * Y_host[i + (size_t)r * j];
            }
            double u11 = std::sqrt(N), u12 = sx / u11, d = sxx - u12 * u12;
            for (int i = 0; i < r; ++i) {
                double b0v, b1v;
                if (!(N > 0.0) || !(d > 0.0)) { b0v = Sy[i]; b1v = sxy[i]; }
                else { double u22 = std::sqrt(d), w1 = Sy[i] / u11, w2 = (sxy[i] - u12 * w1) / u22; b1v = w2 / u22; b0v = (w1 - u12 * b1v) / u11; }
                c0sum[i] += b0v; C[i + r * l] = b1v;
            }
        }
        for (int i = 0; i < r; ++i) C[i] = c0sum[i] / (double)(p + q - 1);
        for (auto &x : C) x = x < -2.0 ? -2.0 : (x > 2.0 ? 2.0 : x);
        C0 = C;
        // project_k!(v): vec(B) with the covariate tail (Inf for kept covariates), top-(k + zkeepn)
        std::vector<double> tail((size_t)r * q);
        for (int l = 0; l < q; ++l) for (int i = 0; i < r; ++i) tail[i + r * l] = zkeep[l] ? std::numeric_limits<double>::infinity() : C[i + r * l];
        hipLaunchKernelGGL(k_mv_full, dim3(nblk(p)), dim3(256), sizeof(double) * 257 * (size_t)r, s, betad.p, p, r, 1.0, full.p, cmat(tail), r * q);
        Sparse snp; std::vector<double> ct; std::vector<uint8_t> cnz;
        MIH_TRY(project_full(snp, ct, cnz));
        B = snp;
        for (int l = 0; l < q; ++l) if (!zkeep[l]) for (int i = 0; i < r; ++i) C[i + r * l] = cnz[i + r * l] ? ct[i + r * l] : 0.0;
        MIH_TRY(choose());
        update_cols();
        for (int l = 0; l < q; ++l) { idc[l] = 0; for (int i = 0; i < r; ++i) if (C[i + r * l] != 0.0) idc[l] = 1; }
        return update_xb();
    }
    // init_iht_indices!(v::mIHTVariable) (multivariate.jl:376-452) = init_pre ; X'R pass ; init_post
    int init(const uint8_t *train)
    {
        MIH_TRY(init_pre(train));
        MIH_TRY(pass());
        return init_post();
    }
    int init_pre(const uint8_t *train)
    {
        if (k < 1) { set_error("Multivariate IHT requires k >= 1!"); return MIH_BAD_ARG; }
        B.clear(); B0.clear(); best_B.clear(); Bg.clear(); B0g.clear(); cols.clear(); dfcols.clear(); spec_ok = false;
        std::fill(C.begin(), C.end(), 0.0); C0 = C; best_C = C; std::fill(df2.begin(), df2.end(), 0.0);
        for (int l = 0; l < q; ++l) idc[l] = zkeep[l];
        std::fill(G.begin(), G.end(), 0.0);
        for (int i = 0; i < r; ++i) G[i + r * i] = 1.0;
        G0 = G;
        choose_fired = false;
        MIH_TRY(set_weights(train, 0));
        nsamples = 0;
        for (int64_t j = 0; j < n; ++j) nsamples += (!train || train[j]);
        if (nsamples == 0) { set_error("no training samples"); return MIH_BAD_ARG; }
        for (int i = 0; i < r; ++i) {                               // intercept = masked trait means (:416-422)
            double ybar = 0.0;
            for (int64_t j = 0; j < n; ++j) if (!train || train[j]) ybar += Y_host[i + (size_t)r * j];
            C[i] = ybar / (double)nsamples;
        }
        MIH_HIP(hipMemsetAsync(BX.p, 0, sizeof(double) * (size_t)r * n, s));
        if (init_beta) MIH_TRY(init_beta_phase(train));
        MIH_TRY(resid_and_gram());                                   // update_mu!, update_resid!
        return score_pre();
    }
    int init_post()
    {
        MIH_TRY(score_post());
        if (init_beta) return step_tail(false);      // the support stays the one project_k!(v) chose; df stays dense
        // vectorize!(full_b, df, df2); project_k!; unvectorize! (:438-440): df replaced by its projection
        std::vector<double> tail((size_t)r * q);
        for (int l = 0; l < q; ++l) for (int i = 0; i < r; ++i) tail[i + r * l] = zkeep[l] ? std::numeric_limits<double>::infinity() : df2[i + r * l];
        hipLaunchKernelGGL(k_mv_full, dim3(nblk(p)), dim3(256), sizeof(double) * 257 * (size_t)r, s, DF.p, p, r, 1.0, full.p, cmat(tail), r * q);
        Sparse snp; std::vector<double> ct; std::vector<uint8_t> cnz;
        MIH_TRY(project_full(snp, ct, cnz));
        if (!comm) hipLaunchKernelGGL(k_mv_unvec, dim3(nblk(p * r)), dim3(256), 0, s, full.p, p, r, DF.p);
        else {               // `full` was not cut at the GLOBAL threshold: df = the survivors, zero elsewhere
            MIH_HIP(hipMemsetAsync(DF.p, 0, sizeof(double) * (size_t)r * p, s));
            MIH_TRY(ensure_stage((int64_t)snp.idx.size()));
            if (!snp.idx.empty()) {
                MIH_TRY(upload_pair(snp.idx.data(), sizeof(int64_t) * snp.idx.size(), sidx.p, snp.val.data(), sizeof(double) * snp.val.size(), sval.p));
                hipLaunchKernelGGL(k_mv_put_df, dim3(nblk((int64_t)snp.idx.size())), dim3(256), 0, s, sidx.p, sval.p, (int64_t)snp.idx.size(), p, r, DF.p);
            }
            Bg.clear();      // (what was projected is the gradient: the model itself is still zero)
        }
        for (int l = 0; l < q; ++l) if (!zkeep[l]) for (int i = 0; i < r; ++i) df2[i + r * l] = cnz[i + r * l] ? ct[i + r * l] : 0.0;
        cols.clear();
        for (size_t t = 0; t < snp.idx.size(); ++t) { int64_t j = snp.idx[t] / r; if (cols.empty() || cols.back() != j) cols.push_back(j); }
        for (int l = 0; l < q; ++l) { idc[l] = 0; for (int i = 0; i < r; ++i) if (df2[i + r * l] != 0.0) idc[l] = 1; }
        return step_tail(false);
    }
    double save_prev(double cur, double best)
    {
        B0 = B; C0 = C; G0 = G; B0g = Bg;
        if (cur > best) { best_B = B; best_C = C; }
        return cur > best ? cur : best;
    }
    int save_best_model()                         // multivariate.jl:485-496: mu includes CZ
    {
        B = best_B; C = best_C;
        MIH_TRY(update_xb());
        hipLaunchKernelGGL(k_mv_resid, dim3(nb), dim3(256), 0, s, Y.p, Z.p, BX.p, w.p, n, r, q, cmat(C), MU.p, RES.p);
        return MIH_OK;
    }
    double check_convergence() const
    {
        const Sparse &B = comm ? this->Bg : this->B, &B0 = comm ? this->B0g : this->B0;      // (a column shard: over the whole model)
        double d = 0.0, nbm = 0.0;
        size_t i = 0, j = 0;
        while (i < B.idx.size() || j < B0.idx.size()) {
            double vb = 0.0, v0 = 0.0;
            if (j >= B0.idx.size() || (i < B.idx.size() && B.idx[i] < B0.idx[j])) vb = B.val[i++];
            else if (i >= B.idx.size() || B0.idx[j] < B.idx[i]) v0 = B0.val[j++];
            else { vb = B.val[i++]; v0 = B0.val[j++]; }
            d = std::max(d, std::fabs(vb - v0)); nbm = std::max(nbm, std::fabs(v0));
        }
        for (size_t t = 0; t < C.size(); ++t) { d = std::max(d, std::fabs(C[t] - C0[t])); nbm = std::max(nbm, std::fabs(C0[t])); }
        return d / (nbm + 1.0);
    }
    int one_step(double old_logl, int nstep, int *bt, double *new_logl)
    {
        MIH_TRY(step_pre(old_logl, nstep, bt, new_logl));
        MIH_TRY(pass());
        return step_post(*new_logl);
    }
    // everything of iht_one_step! before the X'R pass (ends with T1 = Gamma * resid)
    int step_pre(double old_logl, int nstep, int *bt, double *new_logl)
    {
        double eta;
        MIH_TRY(stepsize(&eta));
        MIH_TRY(gradstep(B, C, eta));
        MIH_TRY(update_xb());
        MIH_TRY(solve_sigma());
        double logl = loglik();
        int es = 0;
        while (old_logl > logl && es < nstep) {
            eta /= 2;
            G = G0;                                                  // backtrack! (multivariate.jl:460-473)
            MIH_TRY(gradstep(B0, C0, eta));
            MIH_TRY(update_xb());
            MIH_TRY(solve_sigma());
            logl = loglik();
            es++;
        }
        *bt = es; *new_logl = logl;
        return score_pre();
    }
    int step_post(double logl)
    {
        MIH_TRY(step_tail(true));
        if (std::isnan(logl)) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
        if (std::isinf(logl)) { set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL; }
        return MIH_OK;
    }
    int fit_loop(const mih_fit_params *prm, double *best_out, int64_t *iter_out, double *lt, double *tt, int32_t *btt, int32_t *ntrace)
    {
        double next_logl = -std::numeric_limits<double>::infinity(), best = next_logl;
        int64_t mm = 0; int32_t nt = 0;
        for (int iter = 1; iter <= prm->max_iter; ++iter) {
            if (iter >= prm->max_iter) { best = save_prev(next_logl, best); MIH_TRY(save_best_model()); mm = iter; break; }
            best = save_prev(next_logl, best);
            int nbt = 0;
            // (round 5) check_convergence (multivariate.jl:475-483) reads B, B0, C, C0 only: it is evaluated in front of the X'R pass that
            // ends the step, and a fit that converges here skips that pass -- the reference computes the score inside
            // iht_one_step! and never reads it (26 ms of a 10-trait fit at configs[4])
            MIH_TRY(step_pre(next_logl, prm->max_step, &nbt, &next_logl));
            const double sc = check_convergence();
            if (iter >= prm->min_iter && sc < prm->tol) {
                if (std::isnan(next_logl)) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
                if (std::isinf(next_logl)) { set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL; }
            } else {
                MIH_TRY(pass());
                MIH_TRY(step_post(next_logl));
            }
            if (lt) lt[nt] = next_logl;
            if (tt) tt[nt] = sc;
            if (btt) btt[nt] = nbt;
            nt++;
            if (prm->progress) prm->progress(prm->progress_user, iter, next_logl, nbt, sc);
            if (iter >= prm->min_iter && sc < prm->tol) { best = save_prev(next_logl, best); MIH_TRY(save_best_model()); mm = iter; break; }
        }
        *best_out = best; *iter_out = mm;
        if (ntrace) *ntrace = nt;
        return MIH_OK;
    }
};

static int mv_check(const mih_mat *h, const mih_fit_params *prm)
{
    if (!h || !prm) { set_error("null handle/params"); return MIH_BAD_ARG; }
    if (prm->max_iter < 0 || prm->max_step < 0) { set_error("max_iter / max_step must be nonnegative"); return MIH_BAD_ARG; }
    if (!(prm->tol > 2.220446049250313e-16)) { set_error("Value of global tol must exceed machine precision!"); return MIH_BAD_ARG; }
    if (h->kind == 0 && !h->center) { set_error("x is not centered! Please construct SnpLinAlg{Float64}(::SnpArray, center=true, scale=true)"); return MIH_NOT_CENTERED; }
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

int mih_fit_mv(const mih_mat *h, const mih_fit_params *prm, const double *Y, int64_t r, const double *Z, int64_t q,
               const uint8_t *train, mih_mv_result *res)
{
    PoolScope from_reserve(h ? h->pool : nullptr);      // device buffers out of the matrix's reserve (DevPool, common.h)
    MIH_TRY(mv_check(h, prm));
    if (!Y || !Z || !res) { set_error("null argument"); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(h->device));
    MvVar v;
    MIH_TRY(v.create(h, prm, Y, r, Z, q));
    MIH_TRY(v.init(train));
    auto t0 = std::chrono::steady_clock::now();
    MIH_TRY(v.fit_loop(prm, &res->logl, &res->iter, res->logl_trace, res->tol_trace, res->bt_trace, &res->n_trace));
    res->time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    xtv_count_peels(h, v.xtv, v.s);                  // (measurement hook: how often a trait's row of T1 had rows peeled, csrc/peel.h)
    res->choose_fired = v.choose_fired ? 1 : 0;
    int rr = (int)r;
    if (res->B) {
        std::memset(res->B, 0, sizeof(double) * (size_t)rr * h->p);
        for (size_t t = 0; t < v.best_B.idx.size(); ++t) res->B[v.best_B.idx[t]] = v.best_B.val[t];
    }
    if (res->C) for (size_t t = 0; t < v.best_C.size(); ++t) res->C[t] = v.best_C[t];
    if (res->Sigma) {                                              // inv(v.Gamma) (data_structures.jl:275)
        std::vector<double> inv;
        if (!lu_logdet_inverse(v.G, rr, nullptr, nullptr, &inv)) { set_error("Gamma is singular"); return MIH_NAN_LOGL; }
        for (size_t t = 0; t < inv.size(); ++t) res->Sigma[t] = inv[t];
    }
    if (res->pve) {                                                // pve.jl:36-38
        std::vector<double> mu((size_t)rr * h->n), yy(h->n);
        MIH_HIP(hipMemcpyAsync(mu.data(), v.MU.p, sizeof(double) * mu.size(), hipMemcpyDeviceToHost, v.s));
        MIH_HIP(hipStreamSynchronize(v.s));
        for (int i = 0; i < rr; ++i) {
            for (int64_t s_ = 0; s_ < h->n; ++s_) yy[s_] = Y[i + (size_t)rr * s_];
            res->pve[i] = sample_var(mu.data() + (size_t)i * h->n, h->n) / sample_var(yy.data(), h->n);
        }
    }
    return MIH_OK;
}

int mih_cv_mv(const mih_mat *h, const mih_fit_params *prm, const double *Y, int64_t r, const double *Z, int64_t q,
              const int32_t *folds, int32_t nfolds, const int64_t *path, int64_t npath, int32_t rank, int32_t world,
              double *mses_raw)
{
    PoolScope from_reserve(h ? h->pool : nullptr);      // device buffers out of the matrix's reserve (DevPool, common.h)
    MIH_TRY(mv_check(h, prm));
    if (prm->comm) { set_error("cross-validation shards over (fold,k) combinations (rank/world), not over columns"); return MIH_BAD_ARG; }
    if (!Y || !Z || !folds || !path || !mses_raw || nfolds < 1 || npath < 1 || world < 1 || rank < 0 || rank >= world) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    if (r < 1 || r > kMaxR) { set_error("number of traits r=%lld must be in 1..%d", (long long)r, kMaxR); return MIH_BAD_DIM; }
    int64_t n = h->n;
    for (int64_t i = 0; i < n; ++i) if (folds[i] < 1 || folds[i] > nfolds) { set_error("folds must be in 1..q"); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(h->device));
    for (int64_t i = 0; i < (int64_t)nfolds * npath; ++i) mses_raw[i] = 0.0;
    mih_fit_params pr = *prm;
    pr.progress = nullptr; pr.choose = nullptr;
    // this rank's combinations, fold-major (cross_validation.jl:217-223), advanced in lock-step batches: every
    // round issues ONE fused X'R pass for the r traits of every fit that needs a score (as mih_cv_iht does)
    std::vector<std::pair<int32_t, int64_t>> mine;
    std::vector<int32_t> rank_of;
    cv_assign(path, npath, nfolds, world, rank_of);          // the same sharding rule as mih_cv_iht (fit_lockstep.hip)
    int64_t combo = 0;
    for (int32_t fold = 1; fold <= nfolds; ++fold)
        for (int64_t ik = 0; ik < npath; ++ik, ++combo)
            if (rank_of[(size_t)combo] == rank) mine.emplace_back(fold, ik);
    if (mine.empty()) return MIH_OK;
    const int rr = (int)r;
    const XtvTune tune = xtv_tune(prm);
    const int per_batch = std::max(1, xtv_lockstep_width(h, tune) / rr);
    const int mb = (int)std::min<size_t>(mine.size(), (size_t)per_batch);
    hipStream_t s = nullptr;
    MIH_HIP(hipStreamCreate(&s));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{s};
    XtvWork xw; DevBuf<double> R, DF;
    MIH_TRY(xtv_work_init(h, xw, mb * rr, tune));
    xw.stream_tag = 1;
    MIH_TRY(R.alloc((size_t)mb * rr * n));
    // as in cv_run_rolling (fit_lockstep.hip): the fits of a fold start from the same residuals (initial score before k plays a role,
    // multivariate.jl:376-452) -- one fit per fold rides the pass, its r x p block of X'R is kept behind the pass's outputs;
    // finished fits hand their mIHTVariable (sized for max(path)) to the fits started next
    const bool share_init = !prm->init_beta && probe_env("MENDELIHT_CV_NO_INIT_SHARE") == nullptr;
    const int init_slots = share_init ? (int)std::min<int64_t>(nfolds, 8) : 0;
    MIH_TRY(DF.alloc((size_t)(mb + init_slots) * rr * h->p));
    std::map<int, double *> df0;
    std::vector<std::unique_ptr<MvVar>> pool;
    int64_t kmax = 0;
    for (int64_t i = 0; i < npath; ++i) kmax = std::max(kmax, path[i]);
    struct MvFit {
        std::unique_ptr<MvVar> v; std::vector<uint8_t> train; int64_t out_index = 0; int iter = 1, nbt = 0;
        double next_logl = -std::numeric_limits<double>::infinity(), best = -std::numeric_limits<double>::infinity();
        bool done = false;
        int init_key = -1;
    };
    std::vector<MvFit *> riders;
    std::vector<std::pair<MvFit *, double *>> owners, followers;
    auto batched_xtv = [&](std::vector<MvFit *> &need) -> int {
        const int m = (int)need.size();
        if (m == 0) return MIH_OK;
        for (int t = 0; t < m; ++t)
            MIH_HIP(hipMemcpyAsync(R.p + (size_t)t * rr * n, need[t]->v->T1.p, sizeof(double) * (size_t)rr * n, hipMemcpyDeviceToDevice, s));
        MIH_TRY(xtv_device(h, xw, R.p, m * rr, DF.p, s));
        for (int t = 0; t < m; ++t)
            MIH_HIP(hipMemcpyAsync(need[t]->v->DF.p, DF.p + (size_t)t * rr * h->p, sizeof(double) * (size_t)rr * h->p, hipMemcpyDeviceToDevice, s));
        return MIH_OK;
    };
    auto finish = [&](MvFit &f) -> int {
        MvVar &v = *f.v;
        f.best = v.save_prev(f.next_logl, f.best);
        MIH_TRY(v.save_best_model());
        MIH_TRY(v.set_weights(f.train.data(), 1));
        MIH_TRY(v.update_xb());                               // predict! (cross_validation.jl:288-299)
        hipLaunchKernelGGL(k_mv_resid, dim3(v.nb), dim3(256), 0, v.s, v.Y.p, v.Z.p, v.BX.p, v.w.p, n, v.r, v.q, v.cmat(v.C), v.MU.p, v.RES.p);
        hipLaunchKernelGGL(k_mv_mse, dim3(v.nb), dim3(256), 0, v.s, v.Y.p, v.MU.p, v.w.p, n, v.r, v.red.p);
        hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, v.s, v.red.p, v.nb, 1, v.scal.p);
        double mse = 0.0;
        MIH_HIP(hipMemcpyAsync(&mse, v.scal.p, sizeof(double), hipMemcpyDeviceToHost, v.s));
        MIH_HIP(hipStreamSynchronize(v.s));
        mses_raw[f.out_index] = mse;
        f.done = true;
        pool.push_back(std::move(f.v));
        return MIH_OK;
    };
    // rolling lock-step (as cv_run_rolling in fit_lockstep.hip): a new fit needs its initial score, a running fit the score that
    // ends its step -- one fused pass serves both kinds, and a finished fit's slot is refilled in the next round
    std::vector<std::unique_ptr<MvFit>> slot((size_t)mb);
    std::vector<MvFit *> need;
    std::vector<char> fresh;
    size_t next = 0;
    for (;;) {
        need.clear(); fresh.clear();
        for (int t = 0; t < mb; ++t) {
            for (;;) {
                if (!slot[t]) {
                    if (next >= mine.size()) break;
                    const size_t i = next++;
                    slot[t].reset(new MvFit());
                    MvFit &f = *slot[t];
                    if (!pool.empty()) { f.v = std::move(pool.back()); pool.pop_back(); }
                    else {
                        mih_fit_params pf = pr; pf.k = kmax;                          // buffers sized for max(path), then
                        f.v.reset(new MvVar());
                        MIH_TRY(f.v->create(h, &pf, Y, r, Z, q, s));
                    }
                    f.v->k = path[mine[i].second];                                    // v.k = sparsity (cross_validation.jl:110)
                    f.train.resize(n);
                    for (int64_t l = 0; l < n; ++l) f.train[l] = (folds[l] != mine[i].first);
                    f.out_index = (int64_t)(mine[i].first - 1) * npath + mine[i].second;
                    f.init_key = share_init ? (int)mine[i].first : -1;
                    MIH_TRY(f.v->init_pre(f.train.data()));
                    if (f.init_key >= 0) {
                        auto it = df0.find(f.init_key);
                        if (it != df0.end()) {       // its initial X'R is known from an earlier round: straight on to its first step
                            MIH_HIP(hipMemcpyAsync(f.v->DF.p, it->second, sizeof(double) * (size_t)rr * h->p, hipMemcpyDeviceToDevice, s));
                            MIH_TRY(f.v->init_post());
                            continue;
                        }
                    }
                    need.push_back(&f); fresh.push_back(1);
                    break;
                }
                MvFit &f = *slot[t];
                if (!f.done && f.iter >= pr.max_iter) MIH_TRY(finish(f));                     // fit.jl:170-179
                if (f.done) { slot[t].reset(); continue; }
                f.best = f.v->save_prev(f.next_logl, f.best);
                MIH_TRY(f.v->step_pre(f.next_logl, pr.max_step, &f.nbt, &f.next_logl));
                {   // (round 5) the convergence test needs nothing of the score that ends this step: a fit that converges is finished
                    // without riding the pass, its slot is refilled in this same round (as in the univariate driver, fit_lockstep.hip)
                    const double sc = f.v->check_convergence();
                    if (f.iter >= pr.min_iter && sc < pr.tol) {
                        if (std::isnan(f.next_logl)) { set_error("Loglikelihood function is NaN, aborting..."); return MIH_NAN_LOGL; }
                        if (std::isinf(f.next_logl)) { set_error("Loglikelihood function is Inf, aborting..."); return MIH_INF_LOGL; }
                        MIH_TRY(finish(f));                                               // fit.jl:197-203
                        continue;
                    }
                }
                need.push_back(&f); fresh.push_back(0);
                break;
            }
        }
        if (need.empty()) break;
        riders.clear(); owners.clear(); followers.clear();
        for (size_t t = 0; t < need.size(); ++t) {
            MvFit *f = need[t];
            const int key = fresh[t] ? f->init_key : -1;
            if (key < 0) { riders.push_back(f); continue; }
            auto it = df0.find(key);
            if (it != df0.end()) { followers.emplace_back(f, it->second); continue; }      // same fold, same round: copy the rider's block
            riders.push_back(f);
            if ((int)df0.size() < init_slots) {
                double *buf = DF.p + ((size_t)mb + df0.size()) * (size_t)rr * h->p;
                df0[key] = buf;
                owners.emplace_back(f, buf);
            }
        }
        MIH_TRY(batched_xtv(riders));
        for (auto &o : owners) MIH_HIP(hipMemcpyAsync(o.second, o.first->v->DF.p, sizeof(double) * (size_t)rr * h->p, hipMemcpyDeviceToDevice, s));
        for (auto &fo : followers) MIH_HIP(hipMemcpyAsync(fo.first->v->DF.p, fo.second, sizeof(double) * (size_t)rr * h->p, hipMemcpyDeviceToDevice, s));
        for (size_t t = 0; t < need.size(); ++t) {
            MvFit *f = need[t];
            if (fresh[t]) { MIH_TRY(f->v->init_post()); continue; }
            MIH_TRY(f->v->step_post(f->next_logl));           // (the convergence test of this step ran in front of the pass)
            f->iter++;
        }
    }
    return MIH_OK;
}

}  // extern "C"
