// peel.h -- the outlier side channel of the fixed-point residual (round 6).
//
// X'r runs in fixed point: r is scaled by a power of two so that max|r| fills 54 bits (xtv.hip), every entry is rounded to
// that quantum.  The reference's mul!(df, Transpose(x), r) (src/utilities.jl:133) is a floating-point dot product, so an entry
// r_i keeps its own 53 bits there whatever the largest entry is; here it kept 54 + log2(|r_i| / max|r|) -- with ONE entry 1e8 x
// the rest (a Poisson count the model has not caught up with, an unclamped GLM weight) the columns that do not carry that row
// were good to 2e-7 only.  The side channel: the few rows whose |r_i| towers over the rest are taken OUT of the fixed-point
// residual (their digits are zero, the scale is set by what is left) and their contribution sum_i g_ij r_i is added in plain
// f64 by k_xtv_finalize -- m rows of the 2-bit matrix, read once per column.
//
// The decision is made from exact order statistics, so it does not depend on the order of any sum and is the same in every
// kernel shape, on every rank and in both step modes:
//   * k_r_stats / k_res_stats leave max|r| per strided block of rows (block b: rows 256 b + t + 16384 k): B <= 64 non-empty blocks;
//   * bq = the lower quartile of those block maxima (the ceil(B/4)-th smallest).  Up to ~3/4 of the blocks may hold outliers
//     and bq is still a maximum of ordinary rows;
//   * the guard fires iff max|r| > 64 bq.  Gaussian, Bernoulli, Poisson, log-normal residuals never get there (block maxima of
//     7800 rows and the maximum of 500 000 differ by a factor 1.3 .. 10); then NOTHING changes, not a bit;
//   * if it fires, ONE workgroup counts the rows with |r_i| > tau = 64 bq.  At most kPeelMax of them: they are peeled, listed by
//     ascending row, and the scale comes from the largest |r_i| of the REST (<= tau).  More than kPeelMax (a heavy tail rather
//     than a few outliers): no peel, the plain scale -- the documented 54 + log2(|r_i| / max|r|) bits.
// A peeled residual keeps every entry of the rest to 2^-54 of the rest's maximum and every peeled entry exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mih {

constexpr int kPeelMax = 64;                       // rows the side channel carries per residual
constexpr int kPeelStride = 4 + 2 * kPeelMax;      // doubles per residual: [0] rows peeled (0: none), [1] tau, [2] max |r| of the rest,
                                                   // [3] how often this slot's guard has fired (a running count, for the measurement hook),
                                                   // [4 ..) the rows (as doubles, ascending), [4 + kPeelMax ..) their r_i
constexpr double kPeelRatio = 64.0;

#if defined(__HIPCC__)
// Whole workgroup (blockDim.x a multiple of 64, at most 1024 threads).  bpart[2 b] = max |r| of strided block b (the first
// nb = min(64, ceil(n / 256)) blocks are the non-empty ones).  Writes pl[0 .. 2] (and the lists if rows are peeled) and returns the
// maximum the fixed-point scale is to be taken from: max|r| of the rest if rows were peeled, max|r| otherwise.
__device__ __forceinline__ double peel_decide(const double *__restrict__ r, int64_t n, const double *__restrict__ bpart, int nb,
                                              double *__restrict__ pl)
{
    __shared__ double s_v[64], s_val[kPeelMax], s_red[16];
    __shared__ double s_bq, s_fmx;
    __shared__ int32_t s_row[kPeelMax];
    __shared__ int s_cnt;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid < 64) s_v[tid] = tid < nb ? __hip_atomic_load(&bpart[2 * tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (tid < nb) {                                   // rank of block tid's maximum (ties by block number): exact, whatever the order
        const double mine = s_v[tid];
        int rank = 0;
        for (int c = 0; c < nb; ++c) rank += (s_v[c] < mine || (s_v[c] == mine && c < tid)) ? 1 : 0;
        if (rank == (nb + 3) / 4 - 1) s_bq = mine;
        if (rank == nb - 1) s_fmx = mine;
    }
    __syncthreads();
    const double fmx = nb > 0 ? s_fmx : 0.0, tau = kPeelRatio * (nb > 0 ? s_bq : 0.0);
    if (!(fmx < 1.0e300) || !(fmx > tau)) {           // the common case: no outlier (or a non-finite residual: the scale's own rule)
        if (tid == 0) pl[0] = 0.0;
        return fmx;
    }
    double rest = 0.0;
    for (int64_t i0 = tid; i0 < n; i0 += 8ll * nthr) {
        double x8[8];
        #pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t i = i0 + (int64_t)u * nthr; x8[u] = i < n ? r[i] : 0.0; }
        #pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double a = fabs(x8[u]);
            if (a > tau) {
                const int slot = atomicAdd(&s_cnt, 1);
                if (slot < kPeelMax) { s_row[slot] = (int32_t)(i0 + (int64_t)u * nthr); s_val[slot] = x8[u]; }
            } else rest = fmax(rest, a);
        }
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) rest = fmax(rest, __shfl_xor(rest, off, 64));
    if ((tid & 63) == 0) s_red[tid >> 6] = rest;
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt > kPeelMax) {                             // a heavy tail, not a few outliers: the plain scale
        if (tid == 0) pl[0] = 0.0;
        return fmx;
    }
    rest = 0.0;
    for (int w = 0; w < (nthr + 63) / 64; ++w) rest = fmax(rest, s_red[w]);
    if (tid < cnt) {                                  // ascending rows: the order k_xtv_finalize adds them in
        int rank = 0;
        for (int c = 0; c < cnt; ++c) rank += s_row[c] < s_row[tid] ? 1 : 0;
        pl[4 + rank] = (double)s_row[tid];
        pl[4 + kPeelMax + rank] = s_val[tid];
    }
    if (tid == 0) { pl[0] = (double)cnt; pl[1] = tau; pl[2] = rest; pl[3] += 1.0; }
    return rest;
}
#endif

}  // namespace mih
