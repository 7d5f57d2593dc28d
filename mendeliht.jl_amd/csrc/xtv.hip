// xtv.hip -- out = X' r over the 2-bit genotype matrix: the hot loop of the IHT
// iteration.  Replaces `mul!(v.df, Transpose(x), v.r)` (src/utilities.jl:133;
// SnpArrays.jl linalg_direct.jl `_snparray_AtX_*` kernels) and, batched,
// `SnpArrays.mul!(p_by_r, Transpose(sla), n_by_r)` (src/multivariate.jl:85).
//
//   out_j = sinv_j * ( sum_i g_ij r_i  [+ mu_j * sum_{i missing in j} r_i]  - mu_j * sum_i r_i )
//
// Kernel shape (bandwidth-first, no MFMA): a wave owns C SNP columns and walks the
// rows in chunks of 1024 (64 lanes x one dword = 16 dosages per lane); each
// wave-load is one aligned 256-B segment of a column.  The residual tile for the
// same rows sits in LDS, shared by the workgroup's waves, in a lane-major
// permutation so every ds_read_b128 is conflict-free.  Dosage -> double without
// a convert: the 2-bit field is dropped into the top mantissa bits of 2.0, so
// d = 2 + g/2 and  sum d*r = 2*sum r + (1/2) sum g*r; the constant part is
// removed in the finalize kernel.  Row range can be split over `splits` slices
// (slice = blockIdx % splits, so one XCD keeps re-reading one slice of r from
// its own L2); partials are combined in fixed order => bit-reproducible.
#include "common.h"
#include <mutex>
#include <utility>

namespace mih {

int g_xtv_variant = -1;   // -1: built-in default

// ---- optional per-launch HIP-event timing of the dominant kernel (bench.py roofline) ----
static bool g_profile = false;
static std::mutex g_prof_mu;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
static double g_prof_ms = 0.0;
static int64_t g_prof_launches = 0;

static void prof_begin(hipStream_t s, hipEvent_t &e0, hipEvent_t &e1)
{
    e0 = e1 = nullptr;
    if (!g_profile) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
    (void)hipEventRecord(e0, s);
}
static void prof_end(hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (!e0) return;
    (void)hipEventRecord(e1, s);
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_prof_events.emplace_back(e0, e1);
}

__global__ void k_permute_r(const double *__restrict__ r, int64_t n, int64_t n_perm, int m, int lw,
                            double *__restrict__ rperm)
{
    int64_t total = n_perm * m;
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < total; t += stride) {
        int64_t v = t / n_perm, i = t - v * n_perm;
        rperm[v * n_perm + rperm_pos(i, lw)] = (i < n) ? r[v * n + i] : 0.0;
    }
}

// carrier[v][s] = sum of r over the rows of slice s (fixed order per block => deterministic)
__global__ void __launch_bounds__(256)
k_slice_sums(const double *__restrict__ rperm, int64_t n_perm, int64_t nsc, int rows_per_sc, int splits,
             double *__restrict__ carrier)
{
    __shared__ double red[256];
    int s = blockIdx.x % splits, v = blockIdx.x / splits;
    int64_t sps = (nsc + splits - 1) / splits;
    int64_t c0 = s * sps, c1 = c0 + sps < nsc ? c0 + sps : nsc;
    const double *src = rperm + v * n_perm;
    double a = 0.0;
    for (int64_t i = c0 * rows_per_sc + threadIdx.x; i < c1 * rows_per_sc; i += 256) a += src[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) carrier[v * splits + s] = red[0];
}

// ---- decode + FMA for one dword (16 dosages) -------------------------------------
// MODE 0: bit-field extract + v_cvt_f64_u32 (plain baseline)         acc = sum g r
// MODE 1: shift + and_or into the mantissa of 2.0                    acc = sum (2+g/2) r
// MODE 2: three word shifts + one SDWA byte-select AND per dosage    acc = sum (2+g/2) r
// MODE 3: load-only probe (bench only; results are meaningless)
struct DecodePairs { double d[4]; };   // persistent {lo = 0, hi = 0x40000000 | g << 18} register pairs

template <int MODE>
__device__ __forceinline__ void dot16(uint32_t w, const double (&r)[16], double &acc, DecodePairs &dp);

template <>
__device__ __forceinline__ void dot16<0>(uint32_t w, const double (&r)[16], double &acc, DecodePairs &)
{
    #pragma unroll
    for (int s = 0; s < 16; ++s) acc = fma((double)((w >> (2 * s)) & 3u), r[s], acc);
}

template <>
__device__ __forceinline__ void dot16<1>(uint32_t w, const double (&r)[16], double &acc, DecodePairs &)
{
    #pragma unroll
    for (int s = 0; s < 16; ++s) {
        uint32_t sh = (s <= 9) ? (w << (18 - 2 * s)) : (w >> (2 * s - 18));
        uint32_t hi = (sh & 0x000C0000u) | 0x40000000u;
        acc = fma(__hiloint2double((int)hi, 0), r[s], acc);
    }
}

// The SDWA AND writes (byte K of SRC) & 0x0C into byte 2 of the high word of D and preserves the
// other three bytes, so D keeps {lo = 0, byte 3 = 0x40} for the whole kernel: no re-initialisation.
#define MIH_SDWA_UPD(D, SRC, BYTE)                                                               \
    {                                                                                              \
        uint32_t h_ = (uint32_t)__double2hiint(D);                                                 \
        asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE "                  \
            "src0_sel:BYTE_" #BYTE " src1_sel:DWORD" : "+v"(h_) : "v"(SRC), "v"(mask));            \
        D = __hiloint2double((int)h_, __double2loint(D));                                          \
    }

template <>
__device__ __forceinline__ void dot16<2>(uint32_t w, const double (&r)[16], double &acc, DecodePairs &dp)
{
    // slot q of byte K holds row 4K+q; shift slot q onto bits 2-3 of its byte once per dword.
    const uint32_t mask = 0x0Cu;
    uint32_t w0 = w << 2, w2 = w >> 2, w3 = w >> 4;
#define MIH_BYTE(K)                                                                               \
    MIH_SDWA_UPD(dp.d[0], w0, K) MIH_SDWA_UPD(dp.d[1], w, K)                                      \
    MIH_SDWA_UPD(dp.d[2], w2, K) MIH_SDWA_UPD(dp.d[3], w3, K)                                     \
    acc = fma(dp.d[0], r[4 * K + 0], acc); acc = fma(dp.d[1], r[4 * K + 1], acc);                  \
    acc = fma(dp.d[2], r[4 * K + 2], acc); acc = fma(dp.d[3], r[4 * K + 3], acc);
    MIH_BYTE(0) MIH_BYTE(1) MIH_BYTE(2) MIH_BYTE(3)
#undef MIH_BYTE
}

// Probes (bench only, ids 100+): 3 = loads + LDS + barriers without the decode/FMA work,
// 4 = real decode/FMA + genotype loads but r from registers (no LDS, no staging, no barriers),
// 5 = real decode/FMA + LDS but no genotype loads.
template <>
__device__ __forceinline__ void dot16<3>(uint32_t w, const double (&r)[16], double &acc, DecodePairs &)
{
    acc += __hiloint2double((int)(w & 0x000FFFFFu) | 0x3FF00000, 0) * r[0];
}
template <>
__device__ __forceinline__ void dot16<4>(uint32_t w, const double (&r)[16], double &acc, DecodePairs &dp)
{
    dot16<2>(w, r, acc, dp);
}
template <>
__device__ __forceinline__ void dot16<5>(uint32_t w, const double (&r)[16], double &acc, DecodePairs &dp)
{
    dot16<2>(w, r, acc, dp);
}

__device__ __forceinline__ double wave_sum(double v)
{
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int LW> struct LoadVec;
template <> struct LoadVec<1> { using type = uint32_t; };
template <> struct LoadVec<2> { using type = uint2; };
template <> struct LoadVec<4> { using type = uint4; };
__device__ __forceinline__ uint32_t vec_get(uint32_t v, int) { return v; }
__device__ __forceinline__ uint32_t vec_get(const uint2 &v, int d) { return d == 0 ? v.x : v.y; }
__device__ __forceinline__ uint32_t vec_get(const uint4 &v, int d) { return d == 0 ? v.x : d == 1 ? v.y : d == 2 ? v.z : v.w; }

// WAVES waves x C columns per workgroup.  A lane loads LW consecutive dwords (16*LW rows) of a
// column per "superchunk" of 1024*LW rows, so one wave-load is 256*LW contiguous bytes; TSC
// superchunks of r are staged per LDS buffer (double-buffered).
template <int WAVES, int C, int LW, int TSC, int MODE>
__global__ void __launch_bounds__(WAVES * 64)
k_xtv(const uint32_t *__restrict__ X, int64_t stride_dw, int64_t p,
      const double2 *__restrict__ rperm, int64_t nsc, int splits,
      double *__restrict__ partial /* [splits][p] */)
{
    using Vec = typename LoadVec<LW>::type;
    constexpr int VC = TSC * LW;                  // 1024-row virtual chunks per stage
    __shared__ double2 tile[2][VC * 512];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int split = blockIdx.x % splits;
    const int64_t cg = blockIdx.x / splits;
    const int64_t sps = (nsc + splits - 1) / splits;
    const int64_t c0 = split * sps;
    const int64_t c1 = (c0 + sps < nsc) ? c0 + sps : nsc;
    const int64_t j0 = cg * (WAVES * C) + wave * C;

    const Vec *col[C];
    #pragma unroll
    for (int c = 0; c < C; ++c) {
        int64_t j = j0 + c < p ? j0 + c : p - 1;
        col[c] = reinterpret_cast<const Vec *>(X + j * stride_dw) + lane;
    }
    double acc[C];
    #pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0;
    DecodePairs dp;
    #pragma unroll
    for (int t = 0; t < 4; ++t) dp.d[t] = 2.0;

    // r-tile staging is split (issue the global loads early, store to LDS late) so that the loads'
    // latency hides under the decode/FMA work of the current stage instead of stalling its start.
    constexpr int NST = (VC * 512) / (WAVES * 64);
    static_assert((VC * 512) % (WAVES * 64) == 0, "stage must divide evenly over the workgroup");
    double2 sreg[NST];
    auto stage_load = [&](int64_t cbase) {
        #pragma unroll
        for (int k = 0; k < NST; ++k) {
            int t = threadIdx.x + k * (WAVES * 64);
            int64_t sc = cbase + t / (512 * LW);
            sreg[k] = make_double2(0.0, 0.0);
            if (sc < c1) sreg[k] = rperm[cbase * (512 * LW) + t];
        }
    };
    auto stage_store = [&](int buf) {
        #pragma unroll
        for (int k = 0; k < NST; ++k) tile[buf][threadIdx.x + k * (WAVES * 64)] = sreg[k];
    };

    constexpr bool kUseLds = (MODE != 4);        // probe 4: r from registers, no staging / barriers
    constexpr bool kUseGlobal = (MODE != 5);     // probe 5: no genotype loads
    if (c0 < c1) {
        Vec wcur[C], wnext[C];
        #pragma unroll
        for (int c = 0; c < C; ++c) wcur[c] = col[c][c0 * 64];
        if (kUseLds) {
            stage_load(c0);
            stage_store(0);
            __syncthreads();
        }
        int buf = 0;
        for (int64_t cbase = c0; cbase < c1; cbase += TSC) {
            const bool more = cbase + TSC < c1;
            if (kUseLds && more) stage_load(cbase + TSC);
            const int nsc_here = (c1 - cbase < TSC) ? (int)(c1 - cbase) : TSC;
            for (int sc = 0; sc < nsc_here; ++sc) {
                const int64_t cc = cbase + sc;
                const int64_t cn = (cc + 1 < c1) ? cc + 1 : cc;   // prefetch the next superchunk
                if (kUseGlobal) {
                    #pragma unroll
                    for (int c = 0; c < C; ++c) wnext[c] = col[c][cn * 64];
                }
                #pragma unroll
                for (int d = 0; d < LW; ++d) {
                    double r[16];
                    if (kUseLds) {
                        #pragma unroll
                        for (int m = 0; m < 8; ++m) {
                            double2 v = tile[buf][((sc * LW + d) * 8 + m) * 64 + lane];
                            r[2 * m] = v.x; r[2 * m + 1] = v.y;
                        }
                    } else {
                        #pragma unroll
                        for (int m = 0; m < 16; ++m) r[m] = 1.0 + 0.125 * m + lane;
                    }
                    #pragma unroll
                    for (int c = 0; c < C; ++c) dot16<MODE>(vec_get(wcur[c], d), r, acc[c], dp);
                }
                if (kUseGlobal) {
                    #pragma unroll
                    for (int c = 0; c < C; ++c) wcur[c] = wnext[c];
                } else {
                    #pragma unroll
                    for (int c = 0; c < C; ++c) asm volatile("" : "+v"(wcur[c]));   // keep the decode work alive
                }
            }
            if (kUseLds) {
                if (more) stage_store(buf ^ 1);
                __syncthreads();
            }
            buf ^= 1;
        }
    }
    #pragma unroll
    for (int c = 0; c < C; ++c) {
        double v = wave_sum(acc[c]);
        if (lane == 0 && j0 + c < p) partial[(int64_t)split * p + j0 + c] = v;
    }
}

// Combine slices, undo the mantissa offset, add the missing-entry correction, centre, scale.
__global__ void __launch_bounds__(256)
k_xtv_finalize(const double *__restrict__ partial, const double *__restrict__ carrier, int splits,
               double scaleA, double scaleB, int64_t p, const double *__restrict__ r,
               const double *__restrict__ mu, const double *__restrict__ sinv,
               const int64_t *__restrict__ miss_ptr, const int32_t *__restrict__ miss_row,
               int center, int scale, int impute, double *__restrict__ out)
{
    int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= p) return;
    double dot = 0.0, sumr = 0.0;
    for (int s = 0; s < splits; ++s) {
        dot += scaleA * partial[(int64_t)s * p + j] - scaleB * carrier[s];
        sumr += carrier[s];
    }
    double m = mu[j];
    if (impute) {
        int64_t a = miss_ptr[j], b = miss_ptr[j + 1];
        if (b > a) {
            double ms = 0.0;
            for (int64_t t = a; t < b; ++t) ms += r[miss_row[t]];
            dot += m * ms;
        }
    }
    if (center) dot -= m * sumr;
    if (scale) dot *= sinv[j];
    out[j] = dot;
}

// ---- dense design matrix: out_j = sum_i D[i,j] r_i (one wave per column) ---------------
__global__ void __launch_bounds__(256)
k_xtv_dense(const double *__restrict__ D, int64_t n, int64_t p, const double *__restrict__ r,
            double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    int64_t j = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (j >= p) return;
    const double *col = D + j * n;
    double a0 = 0.0, a1 = 0.0;
    int64_t i = lane * 2;
    if ((n & 1) == 0 && (((uintptr_t)col) & 15) == 0) {
        for (; i + 1 < n; i += 128) {
            double2 x = *reinterpret_cast<const double2 *>(col + i);
            double2 v = *reinterpret_cast<const double2 *>(r + i);
            a0 = fma(x.x, v.x, a0); a1 = fma(x.y, v.y, a1);
        }
    } else {
        for (int64_t k = lane; k < n; k += 64) a0 = fma(col[k], r[k], a0);
    }
    double s = wave_sum(a0 + a1);
    if (lane == 0) out[j] = s;
}

struct Variant { int waves, c, lw, tsc, mode, splits; };
static const Variant kVariants[] = {
    {4, 8, 1, 2, 2, 8},    // 0: SDWA decode, dword loads, 8 XCD-affine row slices
    {4, 8, 1, 2, 1, 8},    // 1: shift+and_or decode
    {4, 8, 1, 2, 0, 8},    // 2: cvt baseline
    {4, 8, 1, 2, 2, 1},    // 3: no row slicing
    {8, 8, 1, 2, 2, 8},    // 4: 8 waves
    {4, 4, 1, 2, 2, 8},    // 5
    {4, 16, 1, 2, 2, 8},   // 6
    {8, 4, 1, 2, 2, 8},    // 7
};
// ids 100+: load-only probes of the same access pattern (bench only; numerically meaningless)
static const Variant kProbes[] = {
    {4, 8, 1, 2, 3, 8},    // 100: loads + LDS, no decode/FMA
    {4, 8, 1, 2, 4, 8},    // 101: loads + decode/FMA, no LDS
    {4, 8, 1, 2, 5, 8},    // 102: LDS + decode/FMA, no genotype loads
    {8, 8, 1, 2, 4, 8},    // 103
    {4, 4, 1, 2, 4, 8},    // 104
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
constexpr int kNumProbes = sizeof(kProbes) / sizeof(kProbes[0]);
constexpr int kDefaultVariant = 0;
int xtv_num_variants() { return kNumVariants; }
bool xtv_variant_valid(int v) { return v < kNumVariants || (v >= 100 && v < 100 + kNumProbes); }

static Variant current_variant()
{
    int v = g_xtv_variant;
    if (v >= 100 && v < 100 + kNumProbes) return kProbes[v - 100];
    if (v < 0 || v >= kNumVariants) v = kDefaultVariant;
    return kVariants[v];
}
int xtv_current_lw() { return current_variant().lw; }

template <int WAVES, int C, int LW, int TSC, int MODE>
static void launch_xtv(const mih_mat *h, const double *rperm, int64_t nsc, int splits,
                       double *partial, hipStream_t s)
{
    int64_t groups = (h->p + WAVES * C - 1) / (WAVES * C);
    dim3 grid((unsigned)(groups * splits));
    hipLaunchKernelGGL((k_xtv<WAVES, C, LW, TSC, MODE>), grid, dim3(WAVES * 64), 0, s, h->X, h->stride_dw, h->p,
                       reinterpret_cast<const double2 *>(rperm), nsc, splits, partial);
}

static int dispatch_xtv(const Variant &v, const mih_mat *h, const double *rperm, int64_t nsc,
                        int splits, double *partial, hipStream_t s)
{
#define MIH_CASE(W, CC, L, T, M) \
    if (v.waves == W && v.c == CC && v.lw == L && v.tsc == T && v.mode == M) { launch_xtv<W, CC, L, T, M>(h, rperm, nsc, splits, partial, s); return MIH_OK; }
    MIH_CASE(4, 8, 1, 2, 2) MIH_CASE(4, 8, 1, 2, 1) MIH_CASE(4, 8, 1, 2, 0) MIH_CASE(8, 8, 1, 2, 2)
    MIH_CASE(4, 4, 1, 2, 2) MIH_CASE(4, 16, 1, 2, 2)
    MIH_CASE(4, 8, 1, 2, 3) MIH_CASE(4, 8, 1, 2, 4) MIH_CASE(4, 8, 1, 2, 5) MIH_CASE(8, 8, 1, 2, 4) MIH_CASE(4, 4, 1, 2, 4) MIH_CASE(8, 4, 1, 2, 2)
#undef MIH_CASE
    set_error("unknown X'r kernel variant");
    return MIH_BAD_ARG;
}

int xtv_work_init(const mih_mat *h, XtvWork &w, int m)
{
    if (h->kind != 0) return MIH_OK;
    int max_splits = 8;
    w.n_perm = h->n_pad;
    MIH_TRY(w.rperm.alloc((size_t)m * (size_t)w.n_perm));
    MIH_TRY(w.partial.alloc((size_t)max_splits * (size_t)m * (size_t)h->p));
    MIH_TRY(w.sums.alloc((size_t)m * (size_t)max_splits));
    w.m_cap = m; w.splits_cap = max_splits;
    return MIH_OK;
}

int xtv_device(const mih_mat *h, XtvWork &w, const double *r_dev, int m, double *out_dev, hipStream_t s)
{
    if (h->kind == 1) {
        for (int v = 0; v < m; ++v)
            hipLaunchKernelGGL(k_xtv_dense, dim3((unsigned)((h->p + 3) / 4)), dim3(256), 0, s, h->D, h->n, h->p,
                               r_dev + (int64_t)v * h->n, out_dev + (int64_t)v * h->p);
        MIH_HIP(hipGetLastError());
        return MIH_OK;
    }
    if (m > w.m_cap) { set_error("X'r workspace too small"); return MIH_BAD_ARG; }
    hipLaunchKernelGGL(k_permute_r, dim3(1024), dim3(256), 0, s, r_dev, h->n, w.n_perm, m, xtv_current_lw(), w.rperm.p);
    return xtv_device_preperm(h, w, r_dev, m, out_dev, s);
}

int xtv_device_preperm(const mih_mat *h, XtvWork &w, const double *r_dev, int m, double *out_dev, hipStream_t s)
{
    Variant v = current_variant();
    int64_t nsc = h->stride_dw / (64 * v.lw);       // superchunks per column
    int splits = v.splits;
    if (splits > nsc) splits = (int)nsc;
    if (splits > w.splits_cap) splits = w.splits_cap;
    double A = (v.mode == 0) ? 1.0 : 2.0, B = (v.mode == 0) ? 0.0 : 4.0;
    hipLaunchKernelGGL(k_slice_sums, dim3((unsigned)(splits * m)), dim3(256), 0, s, w.rperm.p, w.n_perm, nsc, 1024 * v.lw, splits, w.sums.p);
    for (int t = 0; t < m; ++t) {
        double *partial = w.partial.p + (int64_t)t * splits * h->p;
        hipEvent_t e0, e1;
        prof_begin(s, e0, e1);
        int rc = dispatch_xtv(v, h, w.rperm.p + (int64_t)t * w.n_perm, nsc, splits, partial, s);
        prof_end(s, e0, e1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_xtv_finalize, dim3((unsigned)((h->p + 255) / 256)), dim3(256), 0, s, partial,
                           w.sums.p + (int64_t)t * splits, splits, A, B, h->p, r_dev + (int64_t)t * h->n, h->mu, h->sinv,
                           h->miss_ptr, h->miss_row, h->center, h->scale, h->impute, out_dev + (int64_t)t * h->p);
    }
    MIH_HIP(hipGetLastError());
    return MIH_OK;
}

}  // namespace mih

using namespace mih;

extern "C" {

int mih_profile_enable(int on)
{
    g_profile = on != 0;
    return MIH_OK;
}

int mih_profile_read(double *xtv_kernel_ms, int64_t *xtv_launches, int reset)
{
    std::lock_guard<std::mutex> g(g_prof_mu);
    for (auto &pr : g_prof_events) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            g_prof_ms += ms; g_prof_launches++;
        }
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    g_prof_events.clear();
    if (xtv_kernel_ms) *xtv_kernel_ms = g_prof_ms;
    if (xtv_launches) *xtv_launches = g_prof_launches;
    if (reset) { g_prof_ms = 0.0; g_prof_launches = 0; }
    return MIH_OK;
}

int mih_set_xtv_variant(int variant)
{
    if (!xtv_variant_valid(variant)) { set_error("variant %d out of range", variant); return MIH_BAD_ARG; }
    g_xtv_variant = variant;
    return MIH_OK;
}

int mih_xtv_algorithmic_bytes(const mih_mat *h, int m, double *bytes)
{
    if (!h || !bytes || m < 1) return MIH_BAD_ARG;
    if (h->kind == 0)
        *bytes = (double)h->p * (double)((h->n + 3) / 4) + 8.0 * m * ((double)h->n + (double)h->p) + 16.0 * (double)h->p;
    else
        *bytes = 8.0 * (double)h->n * (double)h->p + 8.0 * m * ((double)h->n + (double)h->p);
    return MIH_OK;
}

int mih_xtv_batched(const mih_mat *h, const double *R, int m, double *OUT)
{
    if (!h || !R || !OUT || m < 1) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(h->device));
    XtvWork w;
    MIH_TRY(xtv_work_init(h, w, m));
    DevBuf<double> r, out;
    MIH_TRY(r.alloc((size_t)m * h->n));
    MIH_TRY(out.alloc((size_t)m * h->p));
    MIH_HIP(hipMemcpyAsync(r.p, R, sizeof(double) * (size_t)m * h->n, hipMemcpyHostToDevice, h->stream));
    MIH_TRY(xtv_device(h, w, r.p, m, out.p, h->stream));
    MIH_HIP(hipMemcpyAsync(OUT, out.p, sizeof(double) * (size_t)m * h->p, hipMemcpyDeviceToHost, h->stream));
    MIH_HIP(hipStreamSynchronize(h->stream));
    return MIH_OK;
}

int mih_xtv(const mih_mat *h, const double *r, double *out)
{
    return mih_xtv_batched(h, r, 1, out);
}

__global__ void k_fill_random(double *r, int64_t n, uint64_t seed)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t a = (uint32_t)i * 0x9E3779B1u ^ (uint32_t)seed;
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    uint32_t b = a * 0x85EBCA77u; b ^= b >> 13;
    r[i] = ((a & 0xFFFF) + (a >> 16) + (b & 0xFFFF) + (b >> 16)) * (1.0 / 65536.0) - 2.0;
}

__global__ void k_checksum(const double *x, int64_t p, double *out)
{
    __shared__ double red[256];
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < p; i += 256) a += x[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) *out = red[0];
}

int mih_bench_xtv(const mih_mat *h, int variant, int iters, int warmup, uint64_t seed,
                  float *ms_per_pass, double *checksum)
{
    if (!h || iters < 1 || !ms_per_pass) return MIH_BAD_ARG;
    MIH_HIP(hipSetDevice(h->device));
    int saved = g_xtv_variant;
    if (variant >= 0) { if (!xtv_variant_valid(variant)) return MIH_BAD_ARG; g_xtv_variant = variant; }
    XtvWork w;
    int rc = xtv_work_init(h, w, 1);
    DevBuf<double> r, out, cs;
    if (!rc) rc = r.alloc((size_t)h->n);
    if (!rc) rc = out.alloc((size_t)h->p);
    if (!rc) rc = cs.alloc(1);
    if (rc) { g_xtv_variant = saved; return rc; }
    hipStream_t s = h->stream;
    hipLaunchKernelGGL(k_fill_random, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, s, r.p, h->n, seed);
    for (int i = 0; i < warmup && !rc; ++i) rc = xtv_device(h, w, r.p, 1, out.p, s);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters && !rc; ++i) rc = xtv_device(h, w, r.p, 1, out.p, s);
    (void)hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    g_xtv_variant = saved;
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(e, "bench sync", __FILE__, __LINE__);
    *ms_per_pass = ms / iters;
    if (checksum) {
        hipLaunchKernelGGL(k_checksum, dim3(1), dim3(256), 0, s, out.p, h->p, cs.p);
        MIH_HIP(hipMemcpyAsync(checksum, cs.p, sizeof(double), hipMemcpyDeviceToHost, s));
        MIH_HIP(hipStreamSynchronize(s));
    }
    return MIH_OK;
}

}  // extern "C"
