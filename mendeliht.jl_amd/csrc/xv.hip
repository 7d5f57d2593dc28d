// xv.hip -- out = X[:, S] * v over a small support S (|S| = k << p).
// Replaces the memory-efficient column loops of update_xb! (src/utilities.jl:98-106)
// and iht_stepsize! (src/utilities.jl:731-739), which call SnpLinAlg getindex.
//
// x[i,j] = (g_ij - mu_j) * sinv_j (missing -> mu_j when impute), so
//   out_i = sum_t g_{i,S_t} * a_t + sum_t b_t,  a_t = sinv*v_t,  b_t = -mu*sinv*v_t,
// plus +mu*sinv*v_t on the rows where column S_t is missing (imputed entries are 0
// after centring).  HBM-bound on k columns of 2-bit data (k*ceil(n/4) bytes).
#include "common.h"
#include <algorithm>

namespace mih {

static __global__ void k_xv_gather(const double *__restrict__ src, const int64_t *__restrict__ idx, int64_t nnz, double *__restrict__ out)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < nnz) out[t] = src[idx[t]];
}
// slots_pinned != nullptr: the cache slots of the support sit in pinned host memory (HostStage) and are taken along
__global__ void k_xv_coef(const int64_t *__restrict__ idx, const double *__restrict__ val, int64_t nnz,
                          const double *__restrict__ mu, const double *__restrict__ sinv,
                          int center, int scale, double *__restrict__ A, double *__restrict__ B,
                          const int32_t *__restrict__ slots_pinned = nullptr, int32_t *__restrict__ slots_dev = nullptr,
                          const double *__restrict__ gather_src = nullptr, double *__restrict__ gather_out = nullptr)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    if (slots_pinned) slots_dev[t] = slots_pinned[t];
    int64_t j = idx[t];
    double s = scale ? sinv[j] : 1.0;
    double v = val ? val[t] : gather_src[j];          // the coefficients are gathered on the way (df on the support)
    if (gather_out) gather_out[t] = v;
    double a = s * v;
    A[t] = a;
    B[t] = center ? -mu[j] * a : 0.0;
}

// One workgroup = 16 dwords (256 rows) x 16 column groups: thread (g, d) accumulates the columns of group
// g for the 16 rows of dword d in registers, the 16 group partials meet in LDS and are added in group
// order (fixed order => bit-reproducible), so no n x groups partial array goes through HBM.
constexpr int kXvGroups = 16;
constexpr int kXvPadCols = 16;        // k_xv_snp_cached_mt reads up to the next multiple of eight columns and eight beyond: zero records, repeated offsets
__global__ void __launch_bounds__(256)
k_xv_snp(const uint32_t *__restrict__ X, int64_t nbp, int64_t ndw, int64_t n,
         const int64_t *__restrict__ idx, const double *__restrict__ A, const double *__restrict__ B,
         int64_t nnz, int groups, int clamp20, double *__restrict__ out)
{
    __shared__ double part[kXvGroups][16][17];
    const int g = threadIdx.x >> 4, d = threadIdx.x & 15;
    const int64_t dw = blockIdx.x * 16ll + d;
    const int64_t per = (nnz + groups - 1) / groups;
    const int64_t t0 = g * per, t1 = (g < groups) ? (t0 + per < nnz ? t0 + per : nnz) : t0;
    double acc[16];
    #pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0;
    double bsum = 0.0;
    if (dw < ndw) {
        for (int64_t t = t0; t < t1; ++t) {
            uint32_t w = X[xword(nbp, idx[t], dw)];
            double a = A[t];
            bsum += B[t];
            #pragma unroll
            for (int s = 0; s < 16; ++s) acc[s] = fma((double)((w >> (2 * s)) & 3u), a, acc[s]);
        }
    }
    #pragma unroll
    for (int s = 0; s < 16; ++s) part[g][d][s] = acc[s] + bsum;
    __syncthreads();
    // thread -> (dword dd, row s): sum the groups in order
    const int dd = threadIdx.x >> 4, s = threadIdx.x & 15;
    const int64_t i = (blockIdx.x * 16ll + dd) * 16 + s;
    if (i < n) {
        double a = 0.0;
        for (int gg = 0; gg < groups; ++gg) a += part[gg][dd][s];
        if (clamp20) a = a < -20.0 ? -20.0 : (a > 20.0 ? 20.0 : a);
        out[i] = a;
    }
}

// copy whole columns out of the tile-major matrix into contiguous cache slots; fills[e] = slot << 40 | column
__global__ void __launch_bounds__(256)
k_xv_fill(const uint32_t *__restrict__ X, int64_t nbp, int64_t ndw, const int64_t *__restrict__ fills,
          uint32_t *__restrict__ cache)
{
    int64_t dw = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (dw >= ndw) return;
    const int64_t f = fills[blockIdx.y], slot = f >> 40, col = f & ((1ll << 40) - 1);
    cache[slot * ndw + dw] = X[xword(nbp, col, dw)];
}
// k_xv_snp reading the cached (contiguous) copies of the support columns: same sums in the same order
__global__ void __launch_bounds__(256)
k_xv_snp_cached(const uint32_t *__restrict__ cache, int64_t ndw, int64_t n, const int32_t *__restrict__ slots,
                const double *__restrict__ A, const double *__restrict__ B, int64_t nnz, int groups, int clamp20,
                double *__restrict__ out)
{
    __shared__ double part[kXvGroups][16][17];
    const int g = threadIdx.x >> 4, d = threadIdx.x & 15;
    const int64_t dw = blockIdx.x * 16ll + d;
    const int64_t per = (nnz + groups - 1) / groups;
    const int64_t t0 = g * per, t1 = (g < groups) ? (t0 + per < nnz ? t0 + per : nnz) : t0;
    double acc[16];
    #pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0;
    double bsum = 0.0;
    if (dw < ndw) {
        int64_t t = t0;
        for (; t + 4 <= t1; t += 4) {             // four independent loads in flight (the loop is latency-bound otherwise)
            uint32_t w4[4]; double a4[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) { w4[u] = cache[(int64_t)slots[t + u] * ndw + dw]; a4[u] = A[t + u]; bsum += B[t + u]; }
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                #pragma unroll
                for (int s = 0; s < 16; ++s) acc[s] = fma((double)((w4[u] >> (2 * s)) & 3u), a4[u], acc[s]);
            }
        }
        for (; t < t1; ++t) {
            uint32_t w = cache[(int64_t)slots[t] * ndw + dw];
            double a = A[t];
            bsum += B[t];
            #pragma unroll
            for (int s = 0; s < 16; ++s) acc[s] = fma((double)((w >> (2 * s)) & 3u), a, acc[s]);
        }
    }
    #pragma unroll
    for (int s = 0; s < 16; ++s) part[g][d][s] = acc[s] + bsum;
    __syncthreads();
    const int dd = threadIdx.x >> 4, s = threadIdx.x & 15;
    const int64_t i = (blockIdx.x * 16ll + dd) * 16 + s;
    if (i < n) {
        double a = 0.0;
        for (int gg = 0; gg < groups; ++gg) a += part[gg][dd][s];
        if (clamp20) a = a < -20.0 ? -20.0 : (a > 20.0 ? 20.0 : a);
        out[i] = a;
    }
}

// m coefficient vectors over the same support columns (the traits of a multivariate fit: B[:, idx] * X[idx, :],
// multivariate.jl:21-31, and the step-size products of multivariate.jl:226-236): every cached column is read once per chunk
// of NTR traits instead of once per trait, in ONE launch.  Per trait the sums and their order are those of k_xv_snp_cached.
__global__ void k_xv_coef_multi(const int64_t *__restrict__ idx, const double *__restrict__ val, int64_t nnz,
                                const double *__restrict__ mu, const double *__restrict__ sinv,
                                int center, int scale, double *__restrict__ A, double *__restrict__ B)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    const int64_t o = (int64_t)blockIdx.y * nnz + t, j = idx[t];
    double sc = scale ? sinv[j] : 1.0;
    double a = sc * val[o];
    A[o] = a;
    B[o] = center ? -mu[j] * a : 0.0;
}
template <int NTR>
__global__ void __launch_bounds__(256)
k_xv_snp_cached_multi(const uint32_t *__restrict__ cache, int64_t ndw, int64_t n, const int32_t *__restrict__ slots,
                      const double *__restrict__ A, const double *__restrict__ B, int64_t nnz, int groups, int m,
                      double *__restrict__ out)
{
    __shared__ double part[kXvGroups][16][17];
    const int g = threadIdx.x >> 4, d = threadIdx.x & 15;
    const int64_t dw = blockIdx.x * 16ll + d;
    const int64_t per = (nnz + groups - 1) / groups;
    const int64_t t0 = g * per, t1 = (g < groups) ? (t0 + per < nnz ? t0 + per : nnz) : t0;
    const int v0 = blockIdx.y * NTR;
    double acc[NTR][16], bsum[NTR];
    #pragma unroll
    for (int u = 0; u < NTR; ++u) {
        bsum[u] = 0.0;
        #pragma unroll
        for (int q = 0; q < 16; ++q) acc[u][q] = 0.0;
    }
    if (dw < ndw) {
        const double *Av[NTR], *Bv[NTR];
        #pragma unroll
        for (int u = 0; u < NTR; ++u) { const int v = v0 + u < m ? v0 + u : m - 1; Av[u] = A + (int64_t)v * nnz; Bv[u] = B + (int64_t)v * nnz; }
        int64_t t = t0;
        for (; t + 4 <= t1; t += 4) {             // four independent column loads in flight, the dosages unpacked once per NTR traits
            uint32_t w4[4]; double a4[4][NTR];
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                w4[j] = cache[(int64_t)slots[t + j] * ndw + dw];
                #pragma unroll
                for (int u = 0; u < NTR; ++u) { a4[j][u] = Av[u][t + j]; bsum[u] += Bv[u][t + j]; }
            }
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                #pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const double gq = (double)((w4[j] >> (2 * q)) & 3u);
                    #pragma unroll
                    for (int u = 0; u < NTR; ++u) acc[u][q] = fma(gq, a4[j][u], acc[u][q]);
                }
            }
        }
        for (; t < t1; ++t) {
            const uint32_t w = cache[(int64_t)slots[t] * ndw + dw];
            double a1[NTR];
            #pragma unroll
            for (int u = 0; u < NTR; ++u) { a1[u] = Av[u][t]; bsum[u] += Bv[u][t]; }
            #pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double gq = (double)((w >> (2 * q)) & 3u);
                #pragma unroll
                for (int u = 0; u < NTR; ++u) acc[u][q] = fma(gq, a1[u], acc[u][q]);
            }
        }
    }
    const int dd = threadIdx.x >> 4, q2 = threadIdx.x & 15;
    const int64_t i = (blockIdx.x * 16ll + dd) * 16 + q2;
    #pragma unroll
    for (int u = 0; u < NTR; ++u) {
        #pragma unroll
        for (int q = 0; q < 16; ++q) part[g][d][q] = acc[u][q] + bsum[u];
        __syncthreads();
        if (i < n && v0 + u < m) {
            double a = 0.0;
            for (int gg = 0; gg < groups; ++gg) a += part[gg][dd][q2];
            out[(int64_t)(v0 + u) * n + i] = a;
        }
        __syncthreads();
    }
}

// The same products with a thread per TWO rows and all traits of a chunk in registers: a cached column is unpacked once for
// up to MT traits (the kernel above unpacks it once per 4 traits and, at 256 VGPRs, runs one wave per SIMD), the coefficients
// of a column are wave-uniform (scalar loads), and the thread walks the column groups itself, in order: part_g = acc_g + bsum_g
// joins the total exactly as in the LDS reduction (Bg[v][g] = the group's bsum, summed in column order by k_xv_coef_groups).
// r = 10 traits, 500 columns, n = 500k: 350 us -> ~110 us per call.
template <int MT>
__global__ void __launch_bounds__(256)
k_xv_snp_cached_mt(const uint32_t *__restrict__ cache, int64_t ndw, int64_t n, const int64_t *__restrict__ offs,
                   const double *__restrict__ A, const double *__restrict__ Bg, int nnz, int groups, int m, int rs,
                   double *__restrict__ out)
{
    const int64_t tid = blockIdx.x * 256ll + threadIdx.x;
    const int64_t dw = tid >> 3;
    if (dw >= ndw) return;
    const int sh = (int)(tid & 7) * 4;
    const int v0 = blockIdx.y * MT;
    const int per = (nnz + groups - 1) / groups;
    // (round 4) the coefficients of a column are ONE record of rs = m rounded up to 4 doubles (trait v at A[t * rs + v]): two wide scalar loads
    // and one address per column instead of ten narrow loads with ten addresses, requested one column ahead.  The scalar unit is
    // shared by the CU's four SIMDs and a wave issues one instruction per four cycles whatever its kind: with 30 scalar
    // instructions per column beside the 25 vector ones (counters of tools/pmc_mv.sh: SQ_INSTS_SALU 57.9 M against SQ_INSTS_VALU
    // 56.3 M, the waves 17 % of their time in scalar instructions, 15 % in vector ones) the loop was bound by its bookkeeping.  So:
    // the records and the columns' cache offsets (BYTES, offs[t] = slot * 4 ndw, computed on the host) are PADDED -- records of
    // zeros, the last offset repeated, up to the next multiple of eight and eight beyond -- and the loop neither clamps an
    // index nor tests the column count; pointers advance by a constant.  (A zero record adds +0.0 to sums that are never -0.0.)
    const double *Gv[MT];
    #pragma unroll
    for (int v = 0; v < MT; ++v) { const int vv = v0 + v < m ? v0 + v : m - 1; Gv[v] = Bg + (int64_t)vv * groups; }
    const double *rec = static_cast<const double *>(__builtin_assume_aligned(A + v0, 32));
    // (trait slots past m read the record's padding or the next record: their sums are not stored; the buffer has the slack)
    double acc[MT][2], tot[MT][2];
    #pragma unroll
    for (int v = 0; v < MT; ++v) { acc[v][0] = acc[v][1] = 0.0; tot[v][0] = tot[v][1] = 0.0; }
    int left = per, g = 0;
    // The dwords of a batch of four columns are requested EIGHT columns ahead, into the registers the batch before last has just been
    // taken from (two batches in flight, no copies at the loop's back edge): with four waves per SIMD the 1 - 2 us latency of a
    // batch (the support's 62 MB of columns live in MALL, not in L2) was exposed 125 times per call at 500 columns.
    const char *cbase = reinterpret_cast<const char *>(cache);
    const uint32_t voff = (uint32_t)dw * 4u;
    const int64_t *on = static_cast<const int64_t *>(__builtin_assume_aligned(offs, 64));
    uint32_t wq[2][4];
    #pragma unroll
    for (int u = 0; u < 4; ++u) wq[0][u] = *reinterpret_cast<const uint32_t *>(cbase + on[u] + voff);
    asm volatile("" ::: "memory");          // (the first batch is the older one on entry too: the loop waits with vmcnt(4), never for everything)
    #pragma unroll
    for (int u = 0; u < 4; ++u) wq[1][u] = *reinterpret_cast<const uint32_t *>(cbase + on[4 + u] + voff);
    double a_cur[MT];
    #pragma unroll
    for (int v = 0; v < MT; ++v) a_cur[v] = rec[v];
    for (int t = 0; t < nnz; t += 8) {
        #pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            uint32_t w4[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) { w4[u] = wq[hb][u] >> sh; asm volatile("" : "+v"(w4[u]) :: "memory"); }
            // (the shifted words exist before the requests below are issued: those land in the registers they free, and the loop's
            // back edge needs neither copies nor a wait for everything in flight)
            #pragma unroll
            for (int u = 0; u < 4; ++u) wq[hb][u] = *reinterpret_cast<const uint32_t *>(cbase + on[8 + 4 * hb + u] + voff);
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (left == 0 && t + 4 * hb + u < nnz) {
                    #pragma unroll
                    for (int v = 0; v < MT; ++v) {
                        const double bs = Gv[v][g];
                        tot[v][0] = tot[v][0] + (acc[v][0] + bs); tot[v][1] = tot[v][1] + (acc[v][1] + bs);
                        acc[v][0] = acc[v][1] = 0.0;
                    }
                    ++g; left = per;
                }
                rec = static_cast<const double *>(__builtin_assume_aligned(rec + rs, 32));
                double a_nxt[MT];
                #pragma unroll
                for (int v = 0; v < MT; ++v) a_nxt[v] = rec[v];
                const double g0 = (double)(w4[u] & 3u), g1 = (double)((w4[u] >> 2) & 3u);
                #pragma unroll
                for (int v = 0; v < MT; ++v) {
                    acc[v][0] = fma(g0, a_cur[v], acc[v][0]); acc[v][1] = fma(g1, a_cur[v], acc[v][1]);
                }
                // SMEM returns out of order, so any wait on it is lgkmcnt(0): left to the compiler, that wait lands in FRONT of the next
                // column's multiply-adds and covers the record requested a moment before (the waves 40 % of their time at
                // s_waitcnt).  Waiting here, behind this column's multiply-adds, gives the request their 100 cycles.
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0), vmcnt / expcnt untouched
                #pragma unroll
                for (int v = 0; v < MT; ++v) a_cur[v] = a_nxt[v];
                --left;
            }
        }
        on = static_cast<const int64_t *>(__builtin_assume_aligned(on + 8, 64));
    }
    const double tail = g + 1 < groups ? 0.0 : -0.0;      // empty trailing groups each add +0.0 (only -0.0 + 0.0 differs)
    const int64_t i = 2 * tid;
    #pragma unroll
    for (int v = 0; v < MT; ++v) {
        if (v0 + v < m) {
            const double bs = Gv[v][g];
            const double o0 = (tot[v][0] + (acc[v][0] + bs)) + tail, o1 = (tot[v][1] + (acc[v][1] + bs)) + tail;
            if (i < n) out[(int64_t)(v0 + v) * n + i] = o0;
            if (i + 1 < n) out[(int64_t)(v0 + v) * n + i + 1] = o1;
        }
    }
}
// coefficients of trait blockIdx.x (k_xv_coef_multi's A) and, per column group, the sum of its centring terms in column order
__global__ void __launch_bounds__(256)
k_xv_coef_groups(const int64_t *__restrict__ idx, const double *__restrict__ val, int nnz,
                 const double *__restrict__ mu, const double *__restrict__ sinv, int center, int scale, int groups,
                 double *__restrict__ A, double *__restrict__ Bg, int64_t a_trait_stride, int64_t a_col_stride, int pad_cols)
{
#pragma clang fp contract(off)       // bsum += b must round like the stored B of k_xv_coef_multi
    constexpr int CH = 2048;
    __shared__ double sb[CH];
    const int v = blockIdx.x;
    if ((int)threadIdx.x < pad_cols) A[(int64_t)v * a_trait_stride + (int64_t)(nnz + threadIdx.x) * a_col_stride] = 0.0;    // k_xv_snp_cached_mt's padding records
    const int per = (nnz + groups - 1) / groups;
    const int t0 = threadIdx.x * per, t1 = t0 + per < nnz ? t0 + per : nnz;      // this thread's group (threads < groups)
    double bsum = 0.0;
    for (int base = 0; base < nnz; base += CH) {
        const int end = base + CH < nnz ? base + CH : nnz;
        for (int t = base + threadIdx.x; t < end; t += 256) {
            const int64_t j = idx[t];
            const double sc = scale ? sinv[j] : 1.0;
            const double a = sc * val[(int64_t)v * nnz + t];
            A[(int64_t)v * a_trait_stride + (int64_t)t * a_col_stride] = a;       // trait-major, or one record per column
            sb[t - base] = center ? -mu[j] * a : 0.0;
        }
        __syncthreads();
        if ((int)threadIdx.x < groups)
            for (int t = t0 > base ? t0 : base; t < (t1 < end ? t1 : end); ++t) bsum += sb[t - base];
        __syncthreads();
    }
    if ((int)threadIdx.x < groups) Bg[(int64_t)v * groups + threadIdx.x] = bsum;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_xv_dense(const T *__restrict__ D, int64_t n, const int64_t *__restrict__ idx,
           const double *__restrict__ val, int64_t nnz, int groups, int64_t n_pad,
           double *__restrict__ partial)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int g = blockIdx.y;
    if (i >= n) return;
    int64_t per = (nnz + groups - 1) / groups;
    int64_t t0 = g * per, t1 = t0 + per < nnz ? t0 + per : nnz;
    double acc = 0.0;
    for (int64_t t = t0; t < t1; ++t) acc = fma((double)D[idx[t] * n + i], val[t], acc);
    partial[(int64_t)g * n_pad + i] = acc;
}

__global__ void k_xv_reduce(const double *__restrict__ partial, int groups, int64_t n_pad, int64_t n,
                            int clamp20, double *__restrict__ out)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    for (int g = 0; g < groups; ++g) a += partial[(int64_t)g * n_pad + i];
    if (clamp20) a = a < -20.0 ? -20.0 : (a > 20.0 ? 20.0 : a);
    out[i] = a;
}

// Imputed (missing) entries: one workgroup walks the support columns in order, so the
// additions to any one row happen in a fixed order (bit-reproducible, no atomics).
__global__ void __launch_bounds__(1024)
k_xv_missing(const int64_t *__restrict__ idx, const double *__restrict__ A, int64_t nnz,
             const double *__restrict__ mu,
             const int64_t *__restrict__ miss_ptr, const int32_t *__restrict__ miss_row,
             double *__restrict__ out)
{
    for (int64_t t = 0; t < nnz; ++t) {
        int64_t j = idx[t];
        int64_t a = miss_ptr[j], b = miss_ptr[j + 1];
        double fix = mu[j] * A[t];   // imputed dosage mu_j instead of the stored 0
        for (int64_t e = a + threadIdx.x; e < b; e += blockDim.x) out[miss_row[e]] += fix;
        if (b > a) __syncthreads();
    }
}

__global__ void k_clamp20(double *__restrict__ x, int64_t n)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    x[i] = a < -20.0 ? -20.0 : (a > 20.0 ? 20.0 : a);
}

// device bytes xv_work_init will ask for (the caller sizes its Arena with it)
size_t xv_work_bytes(const mih_mat *h, int64_t max_nnz, int64_t cache_nnz)
{
    if (max_nnz < 64) max_nnz = 64;
    if (cache_nnz <= 0 || cache_nnz > max_nnz) cache_nnz = max_nnz;
    size_t b = 2 * sizeof(double) * (size_t)max_nnz + 4 * 256;
    if (h->kind != 0) return b + sizeof(double) * (size_t)kXvGroups * (size_t)h->n;
    const int64_t ndw = h->n_pad / 16;
    int64_t want = 2 * cache_nnz + 64;
    const int64_t budget = (int64_t)(1ull << 31) / (ndw * 4);
    if (want > budget) want = budget;
    if (want >= cache_nnz) b += (size_t)want * (size_t)ndw * 4 + (size_t)want * 12 + 3 * 256;
    return b;
}

// max_nnz: the longest support list the staging buffers must take (ties included); cache_nnz: the support size the column cache is
// built for (2 x that + 64 slots: a longer list falls back to reading the tile-major matrix directly)
int xv_work_init(const mih_mat *h, XvWork &w, int64_t max_nnz, int64_t cache_nnz)
{
    w.groups = kXvGroups;
    int64_t np = (h->kind == 0) ? h->n_pad : h->n;
    if (h->kind != 0) MIH_TRY(w.partial.alloc((size_t)w.groups * (size_t)np));      // the dense path still reduces through HBM
    if (max_nnz < 64) max_nnz = 64;
    if (cache_nnz <= 0 || cache_nnz > max_nnz) cache_nnz = max_nnz;
    MIH_TRY(w.coefA.alloc((size_t)max_nnz));
    MIH_TRY(w.coefB.alloc((size_t)max_nnz));
    w.cap = max_nnz;
    if (h->kind == 0) {
        const int64_t ndw = h->n_pad / 16;
        int64_t want = 2 * cache_nnz + 64;
        const int64_t budget = (int64_t)(1ull << 31) / (ndw * 4);          // at most 2 GB of cached columns
        if (want > budget) want = budget;
        if (want >= cache_nnz) {
            MIH_TRY(w.cache.alloc((size_t)want * (size_t)ndw));
            MIH_TRY(w.slot_dev.alloc((size_t)want));
            MIH_TRY(w.fill_dev.alloc((size_t)want));
            w.slots = want; w.col_of.assign((size_t)want, -1); w.stamp.assign((size_t)want, 0); w.slot_of.clear(); w.tick = 0;
        }
    }
    return MIH_OK;
}

// map the support columns to cache slots (LRU), queue the copies of the new ones; false: use the direct path
static bool xv_cache_lookup(XvWork &w, const int64_t *idx_host, int64_t nnz, std::vector<int32_t> &slots, std::vector<int64_t> &fills)
{
    if (w.slots <= 0 || nnz > w.slots) return false;
    w.tick++;
    slots.resize((size_t)nnz); fills.clear();
    std::vector<int64_t> miss;
    for (int64_t t = 0; t < nnz; ++t) {
        auto it = w.slot_of.find(idx_host[t]);
        if (it != w.slot_of.end()) { slots[t] = it->second; w.stamp[it->second] = w.tick; }
        else { slots[t] = -1; miss.push_back(t); }
    }
    if (miss.empty()) return true;
    // victims: the least recently used slots not touched by this call
    std::vector<int32_t> order;
    for (int32_t sl = 0; sl < (int32_t)w.slots; ++sl) if (w.stamp[sl] != w.tick) order.push_back(sl);
    if (order.size() < miss.size()) return false;
    std::partial_sort(order.begin(), order.begin() + miss.size(), order.end(),
                      [&](int32_t a, int32_t b) { return w.stamp[a] != w.stamp[b] ? w.stamp[a] < w.stamp[b] : a < b; });
    for (size_t e = 0; e < miss.size(); ++e) {
        const int32_t sl = order[e];
        const int64_t t = miss[e], col = idx_host[t];
        if (w.col_of[sl] >= 0) w.slot_of.erase(w.col_of[sl]);
        w.col_of[sl] = col; w.slot_of[col] = sl; w.stamp[sl] = w.tick;
        slots[t] = sl;
        fills.push_back(((int64_t)sl << 40) | col);
    }
    return true;
}

int xv_sparse_device(const mih_mat *h, XvWork &w, const int64_t *idx_dev, const double *val_dev,
                     int64_t nnz, double *out_dev, int clamp20, hipStream_t s, const int64_t *idx_host, HostStage *st,
                     const double *gather_src, double *gather_out)
{
    if (gather_src && (h->kind != 0 || nnz == 0)) {          // only the 2-bit path gathers inside its coefficient kernel
        if (nnz) hipLaunchKernelGGL(k_xv_gather, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, gather_src, idx_dev, nnz, gather_out);
        val_dev = gather_out; gather_src = nullptr;
    }
    if (nnz == 0) {
        MIH_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * (size_t)h->n, s));
        return MIH_OK;
    }
    if (nnz > w.cap) {
        MIH_HIP(hipStreamSynchronize(s));
        MIH_TRY(w.coefA.alloc((size_t)nnz * 2));
        MIH_TRY(w.coefB.alloc((size_t)nnz * 2));
        w.cap = nnz * 2;
    }
    int groups = (int)(nnz < w.groups ? nnz : w.groups);
    int64_t np = (h->kind == 0) ? h->n_pad : h->n;
    if (h->kind == 0) {
        int64_t ndw = h->n_pad / 16;
        bool fix = h->impute && h->total_missing > 0;
        std::vector<int32_t> &slots = w.h_slots; std::vector<int64_t> &fills = w.h_fills;
        const bool cached = idx_host && h->p < (1ll << 40) && xv_cache_lookup(w, idx_host, nnz, slots, fills);
        // the slot list and the fill list travel through the caller's pinned ring when there is one: k_xv_coef takes the
        // slots along, k_xv_fill reads its (slot, column) pairs there -- no copy operations
        const uint64_t *pin = nullptr;
        if (cached && st) MIH_TRY(st->put(s, slots.data(), sizeof(int32_t) * (size_t)nnz, fills.data(), sizeof(int64_t) * fills.size(), &pin));
        const int32_t *slots_pin = reinterpret_cast<const int32_t *>(pin);
        const int64_t *fills_pin = pin ? reinterpret_cast<const int64_t *>(pin + ((size_t)nnz * sizeof(int32_t) + 7) / 8) : nullptr;
        hipLaunchKernelGGL(k_xv_coef, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, idx_dev, gather_src ? nullptr : val_dev, nnz,
                           h->mu, h->sinv, h->center, h->scale, w.coefA.p, w.coefB.p, slots_pin, w.slot_dev.p, gather_src, gather_out);
        if (gather_src) val_dev = gather_out;
        if (cached) {
            if (!pin) MIH_HIP(hipMemcpyAsync(w.slot_dev.p, slots.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, s));
            if (!fills.empty()) {
                if (!pin) MIH_HIP(hipMemcpyAsync(w.fill_dev.p, fills.data(), sizeof(int64_t) * fills.size(), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_xv_fill, dim3((unsigned)((ndw + 255) / 256), (unsigned)fills.size()), dim3(256), 0, s, h->X, h->nbp, ndw,
                                   pin ? fills_pin : w.fill_dev.p, w.cache.p);
            }
            hipLaunchKernelGGL(k_xv_snp_cached, dim3((unsigned)((ndw + 15) / 16)), dim3(256), 0, s, w.cache.p, ndw, h->n,
                               w.slot_dev.p, w.coefA.p, w.coefB.p, nnz, groups, fix ? 0 : clamp20, out_dev);
        } else
        hipLaunchKernelGGL(k_xv_snp, dim3((unsigned)((ndw + 15) / 16)), dim3(256), 0, s, h->X, h->nbp, ndw, h->n,
                           idx_dev, w.coefA.p, w.coefB.p, nnz, groups, fix ? 0 : clamp20, out_dev);
        if (fix) {
            hipLaunchKernelGGL(k_xv_missing, dim3(1), dim3(1024), 0, s, idx_dev, w.coefA.p, nnz, h->mu, h->miss_ptr, h->miss_row, out_dev);
            if (clamp20) hipLaunchKernelGGL(k_clamp20, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, s, out_dev, h->n);
        }
    } else {
        if (h->Df) hipLaunchKernelGGL(k_xv_dense<float>, dim3((unsigned)((h->n + 255) / 256), groups), dim3(256), 0, s, h->Df, h->n, idx_dev, val_dev,
                                      nnz, groups, np, w.partial.p);
        else hipLaunchKernelGGL(k_xv_dense<double>, dim3((unsigned)((h->n + 255) / 256), groups), dim3(256), 0, s, h->D, h->n, idx_dev, val_dev,
                                nnz, groups, np, w.partial.p);
        hipLaunchKernelGGL(k_xv_reduce, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, s, w.partial.p, groups, np, h->n,
                           clamp20, out_dev);
    }
    MIH_HIP(hipGetLastError());
    return MIH_OK;
}

// out[v*n + i] = sum_t x[i, idx[t]] * vals[v*nnz + t] for v < m (no clamp): one launch over the cached columns when the
// cache holds the support, one single-vector pass per v otherwise.
int xv_sparse_multi_device(const mih_mat *h, XvWork &w, const int64_t *idx_dev, const double *vals_dev, int64_t nnz, int m,
                           double *out_dev, hipStream_t s, const int64_t *idx_host)
{
    const bool fix = h->kind == 0 && h->impute && h->total_missing > 0;
    static const bool multi_on = []() { const char *e = probe_env("MENDELIHT_XV_MULTI"); return !e || atoi(e) != 0; }();
    if (multi_on && m > 1 && nnz > 0 && h->kind == 0 && !fix && idx_host && h->p < (1ll << 40)) {
        if (nnz * m > w.cap) {
            MIH_HIP(hipStreamSynchronize(s));
            MIH_TRY(w.coefA.alloc((size_t)nnz * (m + 3) * 2 + 32));      // (trait-major, or one record of m rounded up to 4 doubles per column)
            MIH_TRY(w.coefB.alloc((size_t)nnz * m * 2));
            w.cap = nnz * m * 2;
        }
        std::vector<int32_t> &slots = w.h_slots; std::vector<int64_t> &fills = w.h_fills;
        if (xv_cache_lookup(w, idx_host, nnz, slots, fills)) {
            const int64_t ndw = h->n_pad / 16;
            const int groups = (int)(nnz < w.groups ? nnz : w.groups);
            static const bool mt_on = []() { const char *e = probe_env("MENDELIHT_XV_MULTI"); return !e || atoi(e) != 2; }();   // 2: the LDS-reduced kernel
            if (mt_on && nnz < (1ll << 30)) {
                if ((size_t)m * groups > w.coefG.n) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(w.coefG.alloc((size_t)m * kXvGroups * 2)); }
                const int rs = (m + 3) & ~3;               // one record per column: the traits side by side (k_xv_snp_cached_mt)
                if ((size_t)(nnz + kXvPadCols) * rs + 32 > w.coefA.n) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(w.coefA.alloc((size_t)(nnz + kXvPadCols) * rs * 2 + 32)); }
                hipLaunchKernelGGL(k_xv_coef_groups, dim3((unsigned)m), dim3(256), 0, s, idx_dev, vals_dev, (int)nnz,
                                   h->mu, h->sinv, h->center, h->scale, groups, w.coefA.p, w.coefG.p, (int64_t)1, (int64_t)rs, kXvPadCols);
            } else
            hipLaunchKernelGGL(k_xv_coef_multi, dim3((unsigned)((nnz + 255) / 256), (unsigned)m), dim3(256), 0, s, idx_dev, vals_dev, nnz,
                               h->mu, h->sinv, h->center, h->scale, w.coefA.p, w.coefB.p);
            const bool mt = mt_on && nnz < (1ll << 30);
            if (mt) {          // byte offsets of the columns' cache slots, the last one repeated over the padding (kXvPadCols)
                std::vector<int64_t> &offs = w.h_offs;
                offs.resize((size_t)(nnz + kXvPadCols));
                for (int64_t t = 0; t < nnz + kXvPadCols; ++t) offs[(size_t)t] = (int64_t)slots[(size_t)(t < nnz ? t : nnz - 1)] * ndw * 4;
                if (offs.size() > w.off_dev.n) { MIH_HIP(hipStreamSynchronize(s)); MIH_TRY(w.off_dev.alloc(offs.size() * 2)); }
                MIH_HIP(hipMemcpyAsync(w.off_dev.p, offs.data(), sizeof(int64_t) * offs.size(), hipMemcpyHostToDevice, s));
            } else
            MIH_HIP(hipMemcpyAsync(w.slot_dev.p, slots.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, s));
            if (!fills.empty()) {
                MIH_HIP(hipMemcpyAsync(w.fill_dev.p, fills.data(), sizeof(int64_t) * fills.size(), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_xv_fill, dim3((unsigned)((ndw + 255) / 256), (unsigned)fills.size()), dim3(256), 0, s, h->X, h->nbp, ndw,
                                   w.fill_dev.p, w.cache.p);
            }
            if (mt) {
                const unsigned gx = (unsigned)((8 * ndw + 255) / 256);
#define MIH_MT(MTV) hipLaunchKernelGGL((k_xv_snp_cached_mt<MTV>), dim3(gx, (unsigned)((m + MTV - 1) / MTV)), dim3(256), 0, s, \
                                       w.cache.p, ndw, h->n, w.off_dev.p, w.coefA.p, w.coefG.p, (int)nnz, groups, m, (m + 3) & ~3, out_dev)
                if (m <= 4) MIH_MT(4); else if (m <= 6) MIH_MT(6); else if (m <= 8) MIH_MT(8); else if (m <= 10) MIH_MT(10); else MIH_MT(12);
#undef MIH_MT
                MIH_HIP(hipGetLastError());
                return MIH_OK;
            }
            constexpr int NTR = 4;
            hipLaunchKernelGGL((k_xv_snp_cached_multi<NTR>), dim3((unsigned)((ndw + 15) / 16), (unsigned)((m + NTR - 1) / NTR)), dim3(256), 0, s,
                               w.cache.p, ndw, h->n, w.slot_dev.p, w.coefA.p, w.coefB.p, nnz, groups, m, out_dev);
            MIH_HIP(hipGetLastError());
            return MIH_OK;
        }
    }
    for (int v = 0; v < m; ++v)
        MIH_TRY(xv_sparse_device(h, w, idx_dev, vals_dev + (size_t)v * nnz, nnz, out_dev + (size_t)v * h->n, 0, s, idx_host));
    return MIH_OK;
}

}  // namespace mih

using namespace mih;

extern "C" int mih_xv_sparse(const mih_mat *h, const int64_t *idx, const double *val, int64_t nnz, double *out)
{
    if (!h || !out || nnz < 0 || (nnz > 0 && (!idx || !val))) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    for (int64_t t = 0; t < nnz; ++t)
        if (idx[t] < 0 || idx[t] >= h->p) { set_error("column index %lld out of range", (long long)idx[t]); return MIH_BAD_DIM; }
    MIH_HIP(hipSetDevice(h->device));
    XvWork w;
    MIH_TRY(xv_work_init(h, w, nnz));
    DevBuf<int64_t> di; DevBuf<double> dv, dout;
    MIH_TRY(di.alloc((size_t)nnz)); MIH_TRY(dv.alloc((size_t)nnz)); MIH_TRY(dout.alloc((size_t)h->n));
    if (nnz) {
        MIH_HIP(hipMemcpyAsync(di.p, idx, sizeof(int64_t) * (size_t)nnz, hipMemcpyHostToDevice, h->stream));
        MIH_HIP(hipMemcpyAsync(dv.p, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, h->stream));
    }
    MIH_TRY(xv_sparse_device(h, w, di.p, dv.p, nnz, dout.p, 0, h->stream));
    MIH_HIP(hipMemcpyAsync(out, dout.p, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    MIH_HIP(hipStreamSynchronize(h->stream));
    return MIH_OK;
}
