// xtv.hip -- out = X' r over the 2-bit genotype matrix: the hot loop of the IHT iteration.
// Replaces `mul!(v.df, Transpose(x), v.r)` (src/utilities.jl:133; SnpArrays.jl linalg_direct.jl
// `_snparray_AtX_*` kernels) and, batched, `SnpArrays.mul!(p_by_r, Transpose(sla), n_by_r)`
// (src/multivariate.jl:85).
//
//   out_j = sinv_j * ( sum_i g_ij r_i  [+ mu_j * sum_{i missing in j} r_i]  - mu_j * sum_i r_i )
//
// Why this is not a plain f64 FMA loop.  Measured on MI355X (tools/instbench.hip): a wave64
// v_fma_f64 costs 2.2 ns per SIMD and the cheapest dosage->double decode (one SDWA byte-select AND)
// 1.85 ns, so a decode+FMA kernel tops out at ~4.0 TB/s of 2-bit data (50 % of the HBM peak) with
// the memory system half idle (the round-1 VALU kernel reached 3.5 TB/s).  The dosage matrix is
// exact small-integer data, so the dot products are done EXACTLY in fixed point on the matrix pipe:
//   * a 2-bit dosage code placed in the low bits of a nibble IS the FP4 (e2m1) number g/2, so the
//     A operand of v_mfma_scale_f32_32x32x64_f8f6f4 is built with one AND per 8 dosages;
//   * r is scaled by a power of two, rounded to an integer R (|R| < 2^54) and written in a positional system
//     whose digits the pipe represents exactly -- each digit set is a complete residue system of its base, so the
//     representation exists, is unique and is found by one pass of divisions per row (DigitMode, k_digits):
//       base  4, digits {-2,-1,0,1}/2                        FP4 e2m1, 28 columns, 1 residual per B operand
//       base 13, digits {-8,-6,-4..4,6,8}/2                  FP4 e2m1, 16 columns (|R| < 2^57), 2 per operand
//       base 49, digits {-32..-18 even, -16..16, 18..32 even}/8  FP6 e2m3, 10 columns, 3 per operand
//     The digit planes of a residual are columns of the 32-column B operand, so ONE MFMA multiplies a
//     32-SNP x 64-row dosage tile with all digits of the operand's residuals.  A single fit uses base 4 (the
//     fastest single-operand pass), every fused multi-RHS context base 49 (fewest MFMAs per residual: those
//     passes are bound by the matrix pipe's power);
//   * every product is a multiple of 1/4 (FP4) or 1/16 (FP6) of magnitude <= 4 and a row slice is short enough
//     (2^22 / 2^20 / 2^18 rows) that every partial sum stays below 2^24 such units: the f32 accumulators are exact
//     and the result does not depend on summation order (bit-reproducible);
//   * the digit sums are recombined in f64 (sum_t base^t * S_t, fixed order) and rescaled by 2^-e.
// The inexact steps are the rounding of r to 2^-54 of max|r| and the few f64 operations that recombine the
// digit sums (fixed order, so bit-reproducible for a given row slicing) -- tighter than the rounding an
// n-term f64 dot product accumulates.  Per 2048 dosages: 1 MFMA (13.7 ns/SIMD) + ~8 VALU ops, far
// below the 87 ns/SIMD the HBM stream allows at 6 TB/s, so the single-operand kernel is memory-bound.
//
// Work decomposition: a wave owns CT column groups (32 SNPs each) and walks a slice of the rows in
// 128-row steps; per step it loads CT x 1 KB of dosages (one contiguous 16 B/lane wave-load per
// tile) and needs the 2 KB of digit planes for those rows.  Rows are split in `splits` slices
// (slice = blockIdx % splits: the blocks of one XCD share a slice of the digit planes in their L2);
// slice partials are combined in fixed order by the finalize kernel.  Two kernel families:
//   k_xtv_mfma_lds<NR,CT,RB,..>  the digit planes of a 128-row block are staged ONCE per workgroup in LDS
//                                (library default for 1, 2 and 4 B operands per pass);
//   k_xtv_mfma<WAVES,CT,NR>      every wave loads its own digit planes from L2 (the earlier shapes, kept as
//                                selectable variants and as a cross-check: same arithmetic, same bits).
// An operand carries one 28-digit base-4 residual, two 16-digit base-13 residuals, three 10-digit base-49 residuals,
// or four 8-digit residuals (opt-in fast modes): DigitMode.
#include "common.h"
#include "peel.h"
#include <algorithm>
#include <mutex>
#include <utility>

namespace mih {

// ---- per-launch HIP-event timing of the dominant kernel, on the matrix handle (mih_profile_*; bench.py roofline) ----
static bool prof_begin(const mih_mat *h, hipStream_t s, PassRecord &rec)
{
    Profile &pf = *h->prof;
    if (!pf.on) return false;
    if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess) { rec.e0 = rec.e1 = nullptr; (void)hipGetLastError(); return false; }
    (void)hipEventRecord(rec.e0, s);
    return true;
}
static void prof_end(const mih_mat *h, hipStream_t s, PassRecord &rec)
{
    (void)hipEventRecord(rec.e1, s);
    Profile &pf = *h->prof;
    std::lock_guard<std::mutex> g(pf.mu);
    pf.open.push_back(rec);
}
void Profile::drain()
{
    for (auto &r : open) {
        float ms = 0.f, st = 0.f;
        mih_pass_record out;
        memset(&out, 0, sizeof(out));
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            if (origin && hipEventElapsedTime(&st, origin, r.e0) != hipSuccess) { st = 0.f; (void)hipGetLastError(); }
            out.start_ms = st; out.ms = ms; out.residuals = r.residuals; out.operands = r.operands; out.stream_tag = r.stream_tag;
            memcpy(out.kernel, r.kernel, sizeof(out.kernel));
            done.push_back(out);
        } else (void)hipGetLastError();
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    open.clear();
}
Profile::~Profile()
{
    for (auto &r : open) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto &r : xopen) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    if (origin) (void)hipEventDestroy(origin);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// ---- residual -> fixed-point digit planes -------------------------------------------------
// scal[4v + 0] = max|r_v|, scal[4v + 1] = 2^-e, scal[4v + 2] = sum r_v, scal[4v + 3] = 2^e
// The block that finishes last (a counter per residual, left at zero again) adds up the partials in block order and
// writes scal: one launch instead of two; the sums do not depend on which block that is.
__global__ void __launch_bounds__(256)
k_r_stats(const double *__restrict__ r, int64_t n, int m, double *__restrict__ part /* [m][gridDim.x][2] */,
          unsigned *__restrict__ done /* [m], zero */, int ebits, double *__restrict__ scal,
          const int32_t *__restrict__ gate, int32_t gate_val, double *__restrict__ peel /* [m][kPeelStride], or null */)
{
    if (gate && *gate != gate_val) return;
    __shared__ double smax[256], ssum[256];
    __shared__ bool last;
    const int v = blockIdx.y;                    // one grid row per residual
    double mx = 0.0, sm = 0.0;
    // eight rows of the thread's walk in flight (the residual is cold after the pass: one load at a time was a round trip per row,
    // 30 of them at n = 500 000); the additions stay in the order of the walk
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 8 * stride) {
        double x8[8];
        #pragma unroll
        for (int u = 0; u < 8; ++u) x8[u] = (i + u * stride < n) ? r[(int64_t)v * n + i + u * stride] : 0.0;
        #pragma unroll
        for (int u = 0; u < 8; ++u) if (i + u * stride < n) { mx = fmax(mx, fabs(x8[u])); sm += x8[u]; }
    }
    smax[threadIdx.x] = mx; ssum[threadIdx.x] = sm;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + k]); ssum[threadIdx.x] += ssum[threadIdx.x + k]; }
        __syncthreads();
    }
    const int nblocks = (int)gridDim.x;
    if (threadIdx.x == 0) {
        part[((int64_t)v * nblocks + blockIdx.x) * 2] = smax[0]; part[((int64_t)v * nblocks + blockIdx.x) * 2 + 1] = ssum[0];
        __threadfence();
        last = atomicAdd(&done[v], 1u) == (unsigned)nblocks - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // (round 6) the outlier side channel (peel.h): the whole block looks at the block maxima; rows that tower over the rest leave the
    // fixed-point residual and the scale is taken from what is left.  No outlier (every residual so far): `top` is max|r|, as before.
    double top = -1.0;
    if (peel) {
        const int64_t nb64 = (n + 255) / 256;
        top = peel_decide(r + (int64_t)v * n, n, part + (int64_t)v * nblocks * 2, (int)(nb64 < nblocks ? nb64 : nblocks), peel + (int64_t)v * kPeelStride);
    }
    if (threadIdx.x == 0) {
        double fmx = 0.0, fsm = 0.0;
        for (int b = 0; b < nblocks; ++b) {
            fmx = fmax(fmx, __hip_atomic_load(&part[((int64_t)v * nblocks + b) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            fsm += __hip_atomic_load(&part[((int64_t)v * nblocks + b) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (top < 0.0) top = fmx;
        // exponent e with top * 2^e < 2^(ebits+1) (DigitMode::ebits); an all-zero (or non-finite) residual keeps e = 0
        int e = 0;
        if (top > 0.0 && top < 1.0e300) e = ebits - ilogb(top);
        if (e > 1000) e = 1000;          // a (numerically zero) residual below 2^-947: keep 2^e finite
        scal[4 * v + 0] = fmx;
        scal[4 * v + 1] = ldexp(1.0, -e);
        scal[4 * v + 2] = fsm;
        scal[4 * v + 3] = ldexp(1.0, e);
        done[v] = 0;
    }
}

// digit of residue m (0..12) in the base-13 system: {-8,-6,-4..4,6,8} is a complete residue system mod 13 whose
// halves are all FP4 (e2m1) numbers; every |R| <= 2^57 has a 16-digit representation (|R| <= 2^27: 8 digits)
__device__ __forceinline__ int digit13(int m)
{
    return m <= 4 ? m : m >= 9 ? m - 13 : m == 5 ? -8 : m == 6 ? 6 : m == 7 ? -6 : 8;
}
// base 49: residues 0..16 and 33..48 (= -16..-1) directly, 17..32 as the even number itself or the odd number
// minus 49 (-32..-18, even): all of them eighths that FP6 (e2m3) represents; |R| <= 2^54 has 10 digits, 2^43 has 8
__device__ __forceinline__ int digit49(int m)
{
    return m <= 16 ? m : m >= 33 ? m - 49 : (m & 1) ? m - 49 : m;
}

// grid (ceil(nblk / 4), residual slots).  One thread per row turns the scaled residual into its digits (FP4 code of
// d/2 or FP6 code of d/8, packed 16 x 4 or 10 x 6 bits per 64-bit word); the 64 rows of a block are then
// transposed through LDS into the B-operand fragment: column sub*slots + t of operand v / per_op holds digit t of
// residual v, lane 32*h + column carries the 32 rows of half h (element order identical to the A fragment built in
// mfma_fp4; an FP6 element j sits in bits 6j..6j+5 of the lane's 192 bits, the last 64 of them in `dig2`).
__global__ void __launch_bounds__(256)
k_digits(const double *__restrict__ r, int64_t n, int64_t nblk, int m, DigitMode dm,
         double *scal, uint4 *__restrict__ dig /* [nops][nblk][64] */, uint2 *__restrict__ dig2, FlatPasses fp, XtvStatsHook sh,
         const double *__restrict__ peel /* [m][kPeelStride]: rows that left the fixed-point residual (peel.h), or null */)
{
    // The wave's 64 rows go into the B-operand image of their block directly: img[wave][half][digit] is the 128 (FP4) or 192 (FP6)
    // bits lane (half, digit) of the fragment carries, and every row ORs its code into its element's place (LDS atomics; round 4 --
    // before, the 2 x slots lanes that own an image gathered it with 32 LDS reads and 64-bit shifts each while the other 44 idled:
    // 60 % of the kernel's time at ten digits).
    if (dm.gate && *dm.gate != dm.gate_val) return;
    __shared__ uint32_t img[4][2][32][6];
    __shared__ double s_scale;
    if (sh.spart) {           // (device-resident steps, one residual) k_r_stats's second stage, by every block for itself; block 0 also leaves it -- and Z'r
        if (threadIdx.x == 0) {
            double fmx = 0.0, fsm = 0.0;
            for (int b = 0; b < 64; ++b) { fmx = fmax(fmx, sh.spart[2 * b]); fsm += sh.spart[2 * b + 1]; }
            const double top = (peel && peel[0] > 0.0) ? peel[2] : fmx;       // (k_res_peel took rows out: the scale of the rest)
            int e = 0;
            if (top > 0.0 && top < 1.0e300) e = sh.ebits - ilogb(top);
            if (e > 1000) e = 1000;
            s_scale = ldexp(1.0, e);
            if (blockIdx.x == 0) { scal[0] = fmx; scal[1] = ldexp(1.0, -e); scal[2] = fsm; scal[3] = ldexp(1.0, e); }
        }
        if (blockIdx.x == 0 && (int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + sh.q) {
            const int l = threadIdx.x - 64;
            double a = 0.0;
            for (int b = 0; b < sh.zblocks; ++b) a += sh.zpart[(int64_t)l * sh.zblocks + b];
            sh.df2[l] = a;
        }
        __syncthreads();
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t blk = blockIdx.x * 4ll + w;
    const int vs = blockIdx.y;
    const int slots = dm.slots, op = vs / dm.per_op, sub = vs % dm.per_op;
    const bool fp6 = dm.base == 49;
    {
        uint32_t *z = &img[w][0][0][0];
        #pragma unroll
        for (int e = 0; e < 6; ++e) z[e * 64 + lane] = 0u;
    }
    __syncthreads();
    if (vs < m && blk < nblk) {
        const int64_t i = blk * 64 + lane;
        double x = (i < n) ? r[(int64_t)vs * n + i] : 0.0;
        if (peel) {                                     // a peeled row's digits are zero: k_xtv_finalize adds its term in f64
            const double *pl = peel + (int64_t)vs * kPeelStride;
            if (pl[0] > 0.0 && fabs(x) > pl[1]) x = 0.0;
        }
        long long R = __double2ll_rn(x * (sh.spart ? s_scale : scal[4 * vs + 3]));
        // element of this row in its half's fragment (the A fragment's order, mfma_fp4): 8 (2u + (s & 1)) + (s >> 1) for row 16u + s
        const int hh = lane >> 5, uu = (lane >> 4) & 1, ss = lane & 15;
        const int el = 8 * (2 * uu + (ss & 1)) + (ss >> 1);
        const int bit = (fp6 ? 6 : 4) * el, dwd = bit >> 5, sh = bit & 31;
        uint32_t *mine = &img[w][hh][0][0];
        // base 49: the standard digits of |R| come from two 5-digit limbs (49^5 < 2^29: 32-bit divisions instead of ten 64-bit
        // ones) and carry into the balanced residue system; the system is symmetric, digits(-R) = -digits(R)
        const bool neg = R < 0;
        const unsigned long long aR = neg ? 0ull - (unsigned long long)R : (unsigned long long)R;
        uint32_t limb_hi = (uint32_t)(aR / 282475249ull), limb_lo = (uint32_t)(aR - (unsigned long long)limb_hi * 282475249ull);
        int carry = 0;
        if (dm.base == 4) {
            // (round 5) all 28 balanced base-4 digits at once: with d_t in {-2,-1,0,1}, R + sum_t 2 * 4^t = sum_t (d_t + 2) 4^t has the
            // ORDINARY base-4 digits u_t = d_t + 2 in {0..3} -- the representation is unique, so these are the digits the division
            // loop below produces -- and the FP4 (e2m1) code of d_t / 2 is a four-entry table: u = 0 -> -1.0 (1010), 1 -> -0.5 (1001),
            // 2 -> 0, 3 -> +0.5 (0001).  No loop-carried dependency.
            const unsigned long long U = (unsigned long long)R + 0x00AAAAAAAAAAAAAAull;
            for (int t = 0; t < dm.ndig; ++t) {
                const uint32_t u = (uint32_t)(U >> (2 * t)) & 3u;
                const uint32_t code = (0x109Au >> (4u * u)) & 15u;
                if (code) atomicOr(mine + t * 6 + dwd, code << sh);
            }
        } else
        for (int t = 0; t < dm.ndig; ++t) {
            int d;
            if (dm.base == 49) {
                const uint32_t L = t < 5 ? limb_lo : limb_hi, qd = L / 49u;
                const int v = (int)(L - qd * 49u) + carry;
                if (t < 5) limb_lo = qd; else limb_hi = qd;
                if (v == 49) { d = 0; carry = 1; }
                else { d = digit49(v); carry = d < 0; }
                if (neg) d = -d;
            } else if (dm.base == 4) {
                const int mm = (int)(R & 3);
                d = mm < 2 ? mm : mm - 4;
                R = (R - d) >> 2;
            } else {
                int mm = (int)(R % 13);
                if (mm < 0) mm += 13;
                d = digit13(mm);
                R = (R - d) / 13;
            }
            const unsigned a = (unsigned)(d < 0 ? -d : d);
            uint32_t code;
            if (fp6) code = (a < 8 ? a : a < 16 ? a : a <= 30 ? 8u + (a >> 1) : 16u + (a >> 2)) | (d < 0 ? 32u : 0u);   // e2m3 of a/8
            else code = (a <= 4 ? a : a == 6 ? 5u : 6u) | (d < 0 ? 8u : 0u);                                          // e2m1 of a/2
            if (code) {
                atomicOr(mine + t * 6 + dwd, code << sh);
                if (sh > 26) atomicOr(mine + t * 6 + dwd + 1, code >> (32 - sh));       // (an FP6 code across a dword boundary)
            }
        }
    }
    __syncthreads();
    if (blk < nblk && lane < 2 * slots) {
        const int h = lane / slots, dg = lane % slots;
        int64_t o = ((int64_t)op * nblk + blk) * 64 + h * 32 + sub * slots + dg;
        if (dm.lay16) {       // image b = column / 16 of the 128-row block, lane 16 * (e + 2h) + column % 16 (k_xtv_dma16)
            int col = sub * slots + dg, opc = op;
            if (dm.flat) {    // digit dg of residual vs sits in column 10 (vs - u0) + dg of its pass, counted across the pass's operands
                int q = 0;
                while (q + 1 < fp.npass && vs >= fp.u0[q + 1]) ++q;
                const int cg = (vs - fp.u0[q]) * slots + dg;
                opc = fp.t0[q] + (cg >> 5); col = cg & 31;
            }
            const int e = (int)(blk & 1);
            o = ((int64_t)opc * nblk + (blk - e) + (col >> 4)) * 64 + 16 * (e + 2 * h) + (col & 15);
        }
        const uint32_t *q = &img[w][h][dg][0];
        dig[o] = make_uint4(q[0], q[1], q[2], q[3]);
        if (fp6) dig2[o] = make_uint2(q[4], q[5]);
    }
}

// ---- the matrix-pipe kernel -------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// dosage tiles are read exactly once per pass: stream them past the caches (nt), so the digit
// planes every wave re-reads stay resident in L2
__device__ __forceinline__ uint4 ld_stream(const uint4 *p)
{
    u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// the same dosage tile against an FP6 (e2m3) B operand: 32 digits x 6 bits = 6 dwords per lane (blgp = 2)
__device__ __forceinline__ f32x16 mfma_fp6(uint32_t u0, uint32_t u1, const uint4 &b, const uint2 &b2, f32x16 acc)
{
    const uint32_t M = 0x33333333u;
    i32x8 a = {(int)(u0 & M), (int)((u0 >> 2) & M), (int)(u1 & M), (int)((u1 >> 2) & M), 0, 0, 0, 0};
    i32x8 bb = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, (int)b2.x, (int)b2.y, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, acc, 4, 2, 0, 0, 0, 0);
}

__device__ __forceinline__ f32x16 mfma_fp4(uint32_t u0, uint32_t u1, const uint4 &b, f32x16 acc)
{
    const uint32_t M = 0x33333333u;
    i32x8 a = {(int)(u0 & M), (int)((u0 >> 2) & M), (int)(u1 & M), (int)((u1 >> 2) & M), 0, 0, 0, 0};
    i32x8 bb = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    // cbsz = blgp = 4: A and B are FP4 (e2m1).  Literal zero scale operands make the compiler select the UNSCALED
    // v_mfma_f32_32x32x64_f8f6f4 (no v_mfma_ld_scale_b32 in front of every MFMA); tools/mfma_probe.hip checks that
    // form against exact integer data
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, acc, 4, 4, 0, 0, 0, 0);
}

// D layout: column n = lane & 31, row (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5) (SNP).  Column n holds digit
// n % slots of residual per_op*v + n / slots of B operand v.
// acc = (1/unit) sum_i g_i d_i exactly (unit 4 for FP4 digits d/2, 16 for FP6 digits d/8); recombine the digits:
// sum_t base^t * (unit * acc_t), then * 2^-e.
template <int CT, int NR>
__device__ __forceinline__ void xtv_epilogue(const f32x16 (&acc)[CT][NR], int lane, int64_t cg0, int64_t ncg, int split,
                                             int splits, DigitMode dm, const double *__restrict__ scal,
                                             double *__restrict__ partial)
{
    const int slots = dm.slots, col = lane & 31;
    const int sub = col / slots, dgt = col - sub * slots;
    const bool live = sub < dm.per_op;                   // columns past per_op * slots carry nothing
    double wgt = 0.0;
    if (live && dgt < dm.ndig) {
        unsigned long long w = dm.base == 49 ? 16 : 4;   // unit * base^dgt < 2^58: exact in 64 bits, one rounding to f64
        for (int t = 0; t < dgt; ++t) w *= (unsigned)dm.base;
        wgt = (double)w;
    }
    const bool tree = (slots & (slots - 1)) == 0;
    const int src0 = (lane & 32) + sub * slots;
    #pragma unroll
    for (int v = 0; v < NR; ++v) {
        const int rhs = v * dm.per_op + (live ? sub : 0);
        const double inv = scal[4 * rhs + 1];
        #pragma unroll
        for (int c = 0; c < CT; ++c) {
            #pragma unroll
            for (int g = 0; g < 16; ++g) {
                double x = (double)acc[c][v][g] * wgt;
                if (tree) {                                                          // fixed tree within each group of
                    if (slots > 16) x += __shfl_xor(x, 16, 64);                      // `slots` lanes
                    if (slots > 8) x += __shfl_xor(x, 8, 64);
                    #pragma unroll
                    for (int off = 4; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
                } else {                                                             // 10 columns: digit 0 upward
                    double sum = 0.0;
                    #pragma unroll
                    for (int t = 0; t < 10; ++t) sum += __shfl(x, src0 + t, 64);
                    x = sum;
                }
                int row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5);
                if (dgt == 0 && live && cg0 + c < ncg)
                    partial[((int64_t)rhs * splits + split) * (ncg * 32) + (cg0 + c) * 32 + row] = x * inv;
            }
        }
    }
}

// NR right-hand sides ride the same pass: the dosage tile is loaded and expanded once and fed to NR
// MFMAs (one per residual vector's digit planes).
template <int WAVES, int CT, int NR>
__global__ void __launch_bounds__(WAVES * 64)
k_xtv_mfma(const uint4 *__restrict__ X, int64_t nbp, int64_t ncg, const uint4 *__restrict__ dig, int64_t dig_stride,
           int splits, DigitMode dm, const double *__restrict__ scal, double *__restrict__ partial /* [NR*per_op][splits][ncg*32] */)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int split = blockIdx.x % splits;
    const int64_t grp = blockIdx.x / splits;
    const int64_t cg0 = (grp * WAVES + wave) * CT;
    if (cg0 >= ncg) return;
    const int64_t bps = (nbp + splits - 1) / splits;
    const int64_t b0 = split * bps;
    const int64_t b1 = (b0 + bps < nbp) ? b0 + bps : nbp;

    f32x16 acc[CT][NR];
    #pragma unroll
    for (int c = 0; c < CT; ++c)
        #pragma unroll
        for (int v = 0; v < NR; ++v)
            #pragma unroll
            for (int g = 0; g < 16; ++g) acc[c][v][g] = 0.f;

    if (b0 < b1) {
        const uint4 *ap[CT];
        #pragma unroll
        for (int c = 0; c < CT; ++c) {
            int64_t cg = cg0 + c < ncg ? cg0 + c : ncg - 1;
            ap[c] = X + (cg * nbp) * 64 + lane;
        }
        const uint4 *bp_ = dig + lane;
        uint4 acur[CT], anext[CT], bcur[NR][2], bnext[NR][2];
        #pragma unroll
        for (int c = 0; c < CT; ++c) acur[c] = ld_stream(ap[c] + b0 * 64);
        #pragma unroll
        for (int v = 0; v < NR; ++v) { bcur[v][0] = bp_[v * dig_stride + (2 * b0) * 64]; bcur[v][1] = bp_[v * dig_stride + (2 * b0 + 1) * 64]; }
        for (int64_t bp = b0; bp < b1; ++bp) {
            const int64_t bn = (bp + 1 < b1) ? bp + 1 : bp;
            #pragma unroll
            for (int c = 0; c < CT; ++c) anext[c] = ld_stream(ap[c] + bn * 64);
            #pragma unroll
            for (int v = 0; v < NR; ++v) { bnext[v][0] = bp_[v * dig_stride + (2 * bn) * 64]; bnext[v][1] = bp_[v * dig_stride + (2 * bn + 1) * 64]; }
            #pragma unroll
            for (int c = 0; c < CT; ++c) {
                #pragma unroll
                for (int v = 0; v < NR; ++v) {
                    acc[c][v] = mfma_fp4(acur[c].x, acur[c].y, bcur[v][0], acc[c][v]);
                    acc[c][v] = mfma_fp4(acur[c].z, acur[c].w, bcur[v][1], acc[c][v]);
                }
            }
            #pragma unroll
            for (int c = 0; c < CT; ++c) acur[c] = anext[c];
            #pragma unroll
            for (int v = 0; v < NR; ++v) { bcur[v][0] = bnext[v][0]; bcur[v][1] = bnext[v][1]; }
        }
    }

    xtv_epilogue<CT, NR>(acc, lane, cg0, ncg, split, splits, dm, scal, partial);
}

// NR (2 or 4) B operands per pass with the digit planes shared through LDS.  A workgroup of WAVES
// waves x CT column groups stages the 8 KB of digit planes of each 128-row block once (instead of once
// per wave: L2 traffic for the digits drops WAVES-fold, which is what keeps the pass off the L2
// roofline) and every wave feeds its dosage tiles to NR x 2 MFMAs per tile.  A barrier step covers RB
// blocks; the digits of step t+1 are loaded during step t-1 and stored to the idle LDS buffer at the
// top of step t, the dosage tiles of step t+1 are loaded at the top of step t, so no load is waited
// for in the step that issued it.  Measured (tools/sweep_multi.py, tools/probe_power.py,
// profiles/r01_power_clock_smi.log): every workgroup shape lands on 29.5 ms because the pass is
// POWER-bound, not issue- or latency-bound -- the package sits at its power cap and the shader clock
// drops to ~1870-1935 MHz for the 4-operand pass; all-zero digit planes run 17 % faster.
// MODE 1 / 2 (no MFMAs / no dosage loads) exist only for those timing probes.
// FP6: the B operands are FP6 digit planes, 24 B per lane: 16 B in `dig` and 8 B in `dig2`, staged side by side.
template <int NR, int CT, int RB, int MODE = 0, int WAVES = 8, bool FP6 = false>   // MODE 1: no MFMAs, 2: no dosage loads (timing probes only)
__global__ void __launch_bounds__(WAVES * 64, 2)
k_xtv_mfma_lds(const uint4 *__restrict__ X, int64_t nbp, int64_t ncg, const uint4 *__restrict__ dig, const uint2 *__restrict__ dig2,
               int64_t dig_stride, int splits, DigitMode dm, const double *__restrict__ scal,
               double *__restrict__ partial /* [NR*per_op][splits][ncg*32] */)
{
    constexpr int NT = WAVES * 64;
    constexpr int BLK = NR * 2 * 64;                 // uint4 slots of one 128-row block: (operand v, 64-row half e, lane)
    constexpr int S = RB * BLK;                      // slots staged per barrier step
    constexpr int PER = (S + NT - 1) / NT;           // slots per thread (the last one may be idle)
    __shared__ uint4 btile[2][S];
    __shared__ uint2 btile2[FP6 ? 2 : 1][FP6 ? S : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int split = blockIdx.x % splits;
    const int64_t grp = blockIdx.x / splits;
    const int64_t cg0 = (grp * WAVES + wave) * CT;
    const int64_t bps = (nbp + splits - 1) / splits;
    const int64_t b0 = split * bps;
    const int64_t b1 = (b0 + bps < nbp) ? b0 + bps : nbp;

    f32x16 acc[CT][NR];
    #pragma unroll
    for (int c = 0; c < CT; ++c)
        #pragma unroll
        for (int v = 0; v < NR; ++v)
            #pragma unroll
            for (int g = 0; g < 16; ++g) acc[c][v][g] = 0.f;

    if (b0 < b1) {
        const int64_t last = b1 - 1;
        const uint4 *ap[CT];
        #pragma unroll
        for (int c = 0; c < CT; ++c) {
            int64_t cg = cg0 + c < ncg ? cg0 + c : ncg - 1;      // idle waves redo the last group
            ap[c] = X + (cg * nbp) * 64 + lane;
        }
        // staged slot f = threadIdx.x + u*NT: block f / BLK of the step, operand (f % BLK) >> 7, half ((f % BLK) >> 6) & 1
        const uint4 *bsrc[PER]; const uint2 *bsrc2[PER]; int bq_[PER]; bool bon[PER];
        #pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int fs = threadIdx.x + u * NT;
            bon[u] = fs < S;
            const int fq = bon[u] ? fs : 0, wi = fq % BLK;
            bq_[u] = fq / BLK;
            const int64_t off = (int64_t)(wi >> 7) * dig_stride + ((wi >> 6) & 1) * 64 + (wi & 63);
            bsrc[u] = dig + off;
            bsrc2[u] = dig2 + off;
        }
        uint4 acur[RB][CT], anext[RB][CT];
        u32x4 bstage[PER];          // native vector type: stays in registers across the loop edge
        u32x2 bstage2[PER];
        #pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int64_t bq = (b0 + q < last) ? b0 + q : last;
            #pragma unroll
            for (int c = 0; c < CT; ++c) acur[q][c] = ld_stream(ap[c] + bq * 64);
        }
        #pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t bq = (b0 + bq_[u] < last) ? b0 + bq_[u] : last;
            const int64_t b2 = (b0 + RB + bq_[u] < last) ? b0 + RB + bq_[u] : last;
            if (bon[u]) btile[0][threadIdx.x + u * NT] = bsrc[u][(2 * bq) * 64];
            bstage[u] = *reinterpret_cast<const u32x4 *>(bsrc[u] + (2 * b2) * 64);
            if (FP6) {
                if (bon[u]) btile2[0][threadIdx.x + u * NT] = bsrc2[u][(2 * bq) * 64];
                bstage2[u] = *reinterpret_cast<const u32x2 *>(bsrc2[u] + (2 * b2) * 64);
            }
        }
        __syncthreads();
        int buf = 0;
        // bstage is carried across the loop edge: the digits of step t+1 are loaded during step t-1
        // and stored to the idle LDS buffer at the top of step t, so neither that load nor the dosage
        // prefetch is waited for in the step that issued it.
        for (int64_t bp = b0; bp < b1; bp += RB) {
            #pragma unroll
            for (int u = 0; u < PER; ++u)
                if (bon[u]) {
                    *reinterpret_cast<u32x4 *>(&btile[buf ^ 1][threadIdx.x + u * NT]) = bstage[u];
                    if (FP6) *reinterpret_cast<u32x2 *>(&btile2[buf ^ 1][threadIdx.x + u * NT]) = bstage2[u];
                }
            #pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int64_t bn = (bp + RB + q < last) ? bp + RB + q : last;
                #pragma unroll
                for (int c = 0; c < CT; ++c) { if (MODE != 2) anext[q][c] = ld_stream(ap[c] + bn * 64); else anext[q][c] = acur[q][c]; }
            }
            #pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int64_t b2 = (bp + 2 * RB + bq_[u] < last) ? bp + 2 * RB + bq_[u] : last;
                bstage[u] = *reinterpret_cast<const u32x4 *>(bsrc[u] + (2 * b2) * 64);
                if (FP6) bstage2[u] = *reinterpret_cast<const u32x2 *>(bsrc2[u] + (2 * b2) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the prefetch loads ahead of the MFMA section
            // (block q, operand v) items in sequence; the digit fragments of item i+1 are read from LDS
            // before the MFMAs of item i are issued so the LDS latency hides behind the matrix pipe
            uint4 bfr[2][2];
            uint2 bfr2[2][2];
            bfr[0][0] = btile[buf][lane];
            bfr[0][1] = btile[buf][64 + lane];
            if (FP6) { bfr2[0][0] = btile2[buf][lane]; bfr2[0][1] = btile2[buf][64 + lane]; }
            #pragma unroll
            for (int i = 0; i < RB * NR; ++i) {
                const int q = i / NR, v = i % NR;
                if (i + 1 < RB * NR) {
                    const int q1 = (i + 1) / NR, v1 = (i + 1) % NR;
                    bfr[(i + 1) & 1][0] = btile[buf][q1 * BLK + (v1 * 2 + 0) * 64 + lane];
                    bfr[(i + 1) & 1][1] = btile[buf][q1 * BLK + (v1 * 2 + 1) * 64 + lane];
                    if (FP6) {
                        bfr2[(i + 1) & 1][0] = btile2[buf][q1 * BLK + (v1 * 2 + 0) * 64 + lane];
                        bfr2[(i + 1) & 1][1] = btile2[buf][q1 * BLK + (v1 * 2 + 1) * 64 + lane];
                    }
                }
                const uint32_t keep = (bp + q < b1) ? 0xFFFFFFFFu : 0u;     // blocks past the slice end add zero
                // the CT tiles against one digit fragment in turn: consecutive MFMAs share the B operand and write
                // different accumulators (2 FP6 operands, CT = 4: 23.1 ms against 23.8 ms for tile-by-tile order)
                if (MODE != 1) {
                    #pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        if (FP6) acc[c][v] = mfma_fp6(acur[q][c].x & keep, acur[q][c].y & keep, bfr[i & 1][0], bfr2[i & 1][0], acc[c][v]);
                        else acc[c][v] = mfma_fp4(acur[q][c].x & keep, acur[q][c].y & keep, bfr[i & 1][0], acc[c][v]);
                    }
                    #pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        if (FP6) acc[c][v] = mfma_fp6(acur[q][c].z & keep, acur[q][c].w & keep, bfr[i & 1][1], bfr2[i & 1][1], acc[c][v]);
                        else acc[c][v] = mfma_fp4(acur[q][c].z & keep, acur[q][c].w & keep, bfr[i & 1][1], acc[c][v]);
                    }
                } else {
                    #pragma unroll
                    for (int c = 0; c < CT; ++c)
                        acc[c][v][0] += __uint_as_float((acur[q][c].x ^ acur[q][c].y ^ acur[q][c].z ^ acur[q][c].w) & keep & bfr[i & 1][0].x & bfr[i & 1][1].y);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            buf ^= 1;
            #pragma unroll
            for (int q = 0; q < RB; ++q)
                #pragma unroll
                for (int c = 0; c < CT; ++c) acur[q][c] = anext[q][c];
        }
    }
    if (cg0 >= ncg) return;
    xtv_epilogue<CT, NR>(acc, lane, cg0, ncg, split, splits, dm, scal, partial);
}

// ---- X'R with every operand through an LDS-DMA ring ---------------------------------------------------------------
// The register-staged LDS kernel above keeps ONE 128-row step of dosage tiles in flight per wave (16 KB per CU with its
// one resident workgroup of the fused shapes) and waits for it at the top of the next step.  Here nothing is loaded
// into registers: every wave copies its own CT dosage tiles and its share of the block's digit planes straight into
// LDS (global_load_lds_dwordx4, 1 KB per instruction) D steps ahead of their use, so D x (WAVES*CT + 2..3 NR) KB are
// in flight per CU and the registers hold only accumulators (AGPRs) and fragments.  Waits are counted by hand
// (s_waitcnt vmcnt(N): LDS-DMA completes in issue order); a wave's own dosage tiles need only its own wait, the shared
// digit planes the wait plus the step's one barrier.  Ring of D + 1 stages: the stage refilled in step t is the one
// last read in step t - 1.  Same arithmetic, same row slicing, same summation order as the other kernels: same bits.
// Measured at n = 500k, p = 1M (tools/sweep_dma.py, tools/probe_dma.py, profiles/r02_*): 12 residuals 34.7 ms against
// 40.0 ms register-staged; D = 2, 3, 4 and the 4 x 4 / 8 x 2 wave shapes all land within 1 % of each other because the
// pass is bound by the package power cap, not by latency or issue (1354 W, shader clock 1.71 GHz; the same MFMAs alone,
// operands in registers, take 20.0 ms at 1.63 GHz: tools/mfma_rate.hip).
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x2v __attribute__((ext_vector_type(2)));
// One ds_read_b64 (64 banks, 2 LDS cycles per wave) that the compiler may not pair with a neighbour: two plain 8-byte loads at
// constant distance become ds_read2_b64 / ds_read2st64_b64, which bank modulo 32 in 16-lane groups -- 8 cycles per instruction,
// and on the A-fragment address map (16-byte lane stride) 2-way conflicts on top: 16 LDS cycles per dosage tile instead of 4
// (round 2's SQ_LDS_BANK_CONFLICT = 8 cycles per tile in every k_xtv_dma16 shape).  Volatile on an LDS-qualified pointer keeps
// the loads apart and in the LDS address space.
typedef __attribute__((address_space(3))) const volatile i32x2v *lds_b64_ptr;
__device__ __forceinline__ i32x2v lds_read_b64(const char *p) { return *(lds_b64_ptr)(p); }

// lane i's 16 B at sbase + voff land at LDS byte address lds_dst + 16 i.  M0 is compiler-reserved: saved and restored.
template <bool NT>
__device__ __forceinline__ void glds16(uint32_t lds_dst, uint32_t voff, const void *sbase)
{
    uint32_t keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}

template <bool FP6>
__device__ __forceinline__ f32x16 mfma4x(i32x4v a, const i32x8 &b, f32x16 acc)
{
    i32x8 aa = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
    if (FP6) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aa, b, acc, 4, 2, 0, 0, 0, 0);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aa, b, acc, 4, 4, 0, 0, 0, 0);
}
// timing probe: consumes the fragments with one VALU operation instead of an MFMA
__device__ __forceinline__ f32x16 fake4x(i32x4v a, const i32x8 &b, f32x16 acc)
{
    acc[0] += __int_as_float((a[0] ^ a[1] ^ a[2] ^ a[3]) & b[0] & b[3]);
    return acc;
}

// Epilogue of the LDS-DMA kernel.  The 32 x 32 accumulator tile of a (column group, operand) goes through a 4.5 KB LDS
// buffer of the wave as f32 [digit column][SNP row] (rows padded to 36 floats: conflict-free 16-B stores); lane
// (row, residual) then adds up its residual's digit columns in exactly the order of xtv_epilogue -- digit 0 upward for
// 10 columns, the xor tree for 8 / 16 / 32 -- so the bits are those of every other kernel, with `slots` LDS reads per
// output instead of 10 f64 shuffles per accumulator register.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int CT, int NR, int SLOTS>
__device__ __forceinline__ void xtv_epilogue_lds_s(const f32x16 (&acc)[CT][NR], float *buf, int lane, int64_t cg0, int64_t ncg,
                                                   int split, int splits, DigitMode dm, const double *__restrict__ scal,
                                                   double *__restrict__ partial)
{
#pragma clang fp contract(off)      // products and sums round separately, as in xtv_epilogue (there a shuffle sits between them)
    constexpr int RS = 36;
    const int col = lane & 31, hi = lane >> 5;
    const int per_op = dm.per_op;
    double wgt[SLOTS];
    {
        unsigned long long w = dm.base == 49 ? 16 : 4;   // unit * base^t < 2^58: exact in 64 bits, one rounding to f64
        #pragma unroll
        for (int t = 0; t < SLOTS; ++t) { wgt[t] = t < dm.ndig ? (double)w : 0.0; w *= (unsigned)dm.base; }
    }
    #pragma unroll
    for (int v = 0; v < NR; ++v) {
        #pragma unroll
        for (int c = 0; c < CT; ++c) {
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4v *>(buf + col * RS + 8 * q + 4 * hi) =
                    f32x4v{acc[c][v][4 * q], acc[c][v][4 * q + 1], acc[c][v][4 * q + 2], acc[c][v][4 * q + 3]};
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int o = lane + 64 * k, row = o & 31, sub = o >> 5;
                if (sub < per_op) {
                    const float *src = buf + sub * SLOTS * RS + row;
                    double x[SLOTS];
                    #pragma unroll
                    for (int t = 0; t < SLOTS; ++t) x[t] = (double)src[t * RS] * wgt[t];
                    double sum;
                    if ((SLOTS & (SLOTS - 1)) == 0) {                // the xor tree of xtv_epilogue, lane 0's cone
                        #pragma unroll
                        for (int off = SLOTS / 2; off > 0; off >>= 1)
                            #pragma unroll
                            for (int t = 0; t < off; ++t) x[t] = x[t] + x[t + off];
                        sum = x[0];
                    } else {
                        sum = 0.0;
                        #pragma unroll
                        for (int t = 0; t < SLOTS; ++t) sum += x[t];
                    }
                    const int rhs = v * per_op + sub;
                    if (cg0 + c < ncg)
                        partial[((int64_t)rhs * splits + split) * (ncg * 32) + (cg0 + c) * 32 + row] = sum * scal[4 * rhs + 1];
                }
            }
        }
    }
}
template <int CT, int NR>
__device__ __forceinline__ void xtv_epilogue_lds(const f32x16 (&acc)[CT][NR], float *buf, int lane, int64_t cg0, int64_t ncg,
                                                 int split, int splits, DigitMode dm, const double *__restrict__ scal,
                                                 double *__restrict__ partial)
{
    if (dm.slots == 10) xtv_epilogue_lds_s<CT, NR, 10>(acc, buf, lane, cg0, ncg, split, splits, dm, scal, partial);
    else if (dm.slots == 8) xtv_epilogue_lds_s<CT, NR, 8>(acc, buf, lane, cg0, ncg, split, splits, dm, scal, partial);
    else if (dm.slots == 16) xtv_epilogue_lds_s<CT, NR, 16>(acc, buf, lane, cg0, ncg, split, splits, dm, scal, partial);
    else xtv_epilogue_lds_s<CT, NR, 32>(acc, buf, lane, cg0, ncg, split, splits, dm, scal, partial);
}

template <int NR, int CT, int WAVES, int D, bool FP6, int MODE = 0>      // MODE 1: no MFMAs, 2: no copies after the prologue (timing probes only)
__global__ void __launch_bounds__(WAVES * 64, 1)
k_xtv_dma(const uint4 *__restrict__ X, int64_t nbp, int64_t ncg, const uint4 *__restrict__ dig, const uint2 *__restrict__ dig2,
          int64_t dig_stride, int splits, DigitMode dm, const double *__restrict__ scal,
          double *__restrict__ partial /* [NR*per_op][splits][ncg*32] */)
{
    constexpr int S = D + 1;                        // ring stages
    constexpr int DOS = WAVES * CT * 1024;          // dosage bytes of a stage (wave w: tiles at w*CT KB)
    constexpr int OPB = FP6 ? 3072 : 2048;          // digit bytes of an operand and step: 2 x 1 KB `dig` halves (+ 1 KB `dig2`)
    constexpr int DGT = NR * OPB;
    constexpr int STAGE = DOS + DGT;
    constexpr int PPO = FP6 ? 3 : 2;                // 1 KB pieces per operand
    constexpr int NP = NR * PPO;                    // digit pieces per step
    constexpr int PW = (NP + WAVES - 1) / WAVES;    // pieces per wave (surplus slots repeat the last piece)
    constexpr int L = CT + PW;                      // LDS-DMA instructions per wave and step
    constexpr int NI = 2 * NR;                      // (operand, 64-row half) items of a step
    static_assert(S * STAGE <= 160 * 1024, "LDS ring too large");
    static_assert(D * L <= 63, "vmcnt range");
    static_assert(S * STAGE >= WAVES * 32 * 36 * 4, "the epilogue buffers overlay the ring");
    if (dm.gate && *dm.gate != dm.gate_val) return;      // (uniform: one scalar load and a branch in front of everything)
    __shared__ uint4 lds[S * STAGE / 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int split = blockIdx.x % splits;
    const int64_t grp = blockIdx.x / splits;
    const int64_t cg0 = (grp * WAVES + wave) * CT;
    const int64_t bps = (nbp + splits - 1) / splits;
    const int64_t b0 = split * bps;
    const int64_t b1 = (b0 + bps < nbp) ? b0 + bps : nbp;

    f32x16 acc[CT][NR];
    #pragma unroll
    for (int c = 0; c < CT; ++c)
        #pragma unroll
        for (int v = 0; v < NR; ++v)
            #pragma unroll
            for (int g = 0; g < 16; ++g) acc[c][v][g] = 0.f;

    // (an empty trailing slice -- nbp not a multiple of the slice count -- runs one clamped, masked-out step so that
    // the accumulators never live across a branch: that would push all of them through scratch)
    {
        const bool empty = b0 >= b1;
        const int64_t bb0 = empty ? nbp - 1 : b0;
        const int nb = empty ? 1 : (int)(b1 - b0);
        const int amask = empty ? 0 : -1;
        const char *xs[CT];
        #pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int64_t cg = cg0 + c < ncg ? cg0 + c : ncg - 1;        // idle waves redo the last group
            xs[c] = reinterpret_cast<const char *>(X) + (cg * nbp + bb0) * 1024;
        }
        const char *dsrc[PW]; int dstep[PW]; int doff[PW];
        #pragma unroll
        for (int u = 0; u < PW; ++u) {
            const int jj = wave + u * WAVES, j = jj < NP ? jj : NP - 1;
            const int op = j / PPO, part = j - PPO * op;
            if (part < 2) {
                dsrc[u] = reinterpret_cast<const char *>(dig) + (op * dig_stride + (2 * bb0 + part) * 64) * 16;
                dstep[u] = 2048; doff[u] = DOS + op * OPB + part * 1024;
            } else {
                dsrc[u] = reinterpret_cast<const char *>(dig2) + (op * dig_stride + 2 * bb0 * 64) * 8;
                dstep[u] = 1024; doff[u] = DOS + op * OPB + 2048;
            }
        }
        const uint32_t voff = lane * 16;
        const uint32_t lds0 = (uint32_t)(uintptr_t)lds;           // low 32 bits of a generic LDS address = the byte offset M0 takes
        const uint32_t mydos = wave * CT * 1024;
        const char *ldsb = reinterpret_cast<const char *>(lds);

        auto issue = [&](int ts, int st) {
            const int tb = ts < nb ? ts : nb - 1;                        // past the slice end: copy the last block again
            const uint32_t base = lds0 + st * STAGE;
            #pragma unroll
            for (int c = 0; c < CT; ++c) glds16<true>(base + mydos + c * 1024, voff, xs[c] + (int64_t)tb * 1024);
            #pragma unroll
            for (int u = 0; u < PW; ++u) glds16<false>(base + doff[u], voff, dsrc[u] + (int64_t)tb * dstep[u]);
        };
        auto read_b = [&](int st, int item, i32x8 &b) {
            const int v = item >> 1, e = item & 1;
            const char *q = ldsb + st * STAGE + DOS + v * OPB;
            const i32x4v lo = *reinterpret_cast<const i32x4v *>(q + e * 1024 + lane * 16);
            if (FP6) {
                const i32x2v hi = lds_read_b64(q + 2048 + e * 512 + lane * 8);
                b = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], 0, 0};
            } else b = i32x8{lo[0], lo[1], lo[2], lo[3], 0, 0, 0, 0};
        };
        auto read_dos = [&](int st, i32x4v (&raw)[CT]) {
            #pragma unroll
            for (int c = 0; c < CT; ++c) raw[c] = *reinterpret_cast<const i32x4v *>(ldsb + st * STAGE + mydos + c * 1024 + lane * 16);
        };
        auto expand = [&](const i32x4v (&raw)[CT], i32x4v (&a)[CT][2]) {
            const int M = 0x33333333 & amask;
            #pragma unroll
            for (int c = 0; c < CT; ++c) {
                const unsigned x = raw[c][0], y = raw[c][1], z = raw[c][2], w = raw[c][3];
                a[c][0] = i32x4v{(int)x & M, (int)(x >> 2) & M, (int)y & M, (int)(y >> 2) & M};
                a[c][1] = i32x4v{(int)z & M, (int)(z >> 2) & M, (int)w & M, (int)(w >> 2) & M};
            }
        };

        #pragma unroll
        for (int s = 0; s < D; ++s) issue(s, s);
        i32x4v araw[CT];
        i32x4v A[2][CT][2];
        i32x8 B[2];
        wait_vm_barrier<(D - 1) * L>();
        read_dos(0, araw);
        read_b(0, 0, B[0]);
        expand(araw, A[0]);
        int st = 0;
        // one step: the MFMAs of step T with the fragments of buffer P; fills buffer P ^ 1 for step T + 1.  The last item's
        // MFMAs are issued after the barrier, behind the first fragment read of the next step.
#define MIH_MM(a_, b_, c_) (MODE == 1 ? fake4x(a_, b_, c_) : mfma4x<FP6>(a_, b_, c_))
#define MIH_DMA_STEP(P, T)                                                                                         \
        {                                                                                                          \
            const int st_next = st + 1 == S ? 0 : st + 1, st_ld = st == 0 ? S - 1 : st - 1;                        \
            if (MODE != 2) issue((T) + D, st_ld);                                                                  \
            _Pragma("unroll")                                                                                      \
            for (int i = 0; i < NI - 1; ++i) {                                                                     \
                read_b(st, i + 1, B[(i + 1) & 1]);                                                                 \
                _Pragma("unroll")                                                                                  \
                for (int c = 0; c < CT; ++c) acc[c][i >> 1] = MIH_MM(A[P][c][i & 1], B[i & 1], acc[c][i >> 1]);    \
                if (i == (NI > 2 ? NI / 2 - 1 : 0)) { wait_vm<D * L - CT>(); read_dos(st_next, araw); }            \
                if (i == (NI > 2 ? NI / 2 : 0)) expand(araw, A[(P) ^ 1]);                                          \
            }                                                                                                      \
            wait_vm_barrier<(D - 1) * L>();                                                                        \
            read_b(st_next, 0, B[0]);                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            _Pragma("unroll")                                                                                      \
            for (int c = 0; c < CT; ++c) acc[c][NR - 1] = MIH_MM(A[P][c][1], B[1], acc[c][NR - 1]);                \
            st = st_next;                                                                                          \
        }
        for (int t = 0; t < nb; t += 2) {
            MIH_DMA_STEP(0, t)
            if (t + 1 < nb) MIH_DMA_STEP(1, t + 1)
        }
#undef MIH_DMA_STEP
#undef MIH_MM
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the look-ahead copies have landed, every wave is done with the ring
    }
    if (cg0 >= ncg) return;
    xtv_epilogue_lds<CT, NR>(acc, reinterpret_cast<float *>(lds) + wave * (32 * 36), lane, cg0, ncg, split, splits, dm, scal, partial);
}

// The same pass on v_mfma_f32_16x16x128_f8f6f4.  Under the package power cap the chip holds a higher clock on the
// 16x16x128 form of the instruction (half the accumulator traffic per multiply-add): the same multiply-adds, operands in
// registers, take 16.0 ms against 20.0 ms for 32x32x64 on dosage-like x digit-like data (tools/mfma_rate.hip).  A tile
// (32 SNPs x 128 rows) becomes two A fragments (16 SNPs each, all 128 rows: lane (r, kq) reads the 8 bytes of row group
// (e, h) = (kq & 1, kq >> 1) of SNP r from the LDS image -- the row-group order is free as long as both operands use it),
// an operand's digit planes two B fragments of 16 columns (DigitMode::lay16 layout written by k_digits), and the four
// 16 x 16 products of a (tile, operand) accumulate over the whole 128-row block in one instruction each.
typedef float f32x4a __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4a mfma16(i32x4v a, const i32x8 &b, f32x4a acc)
{
    i32x8 aa = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa, b, acc, 4, 2, 0, 0, 0, 0);
}

template <int CT, int NR, int SLOTS, int HALF>
__device__ __forceinline__ void xtv_epilogue16_s(const f32x4a (&acc)[CT][NR][2][2], float *buf, int lane, int64_t cg0, int64_t ncg,
                                                 int split, int splits, DigitMode dm, const double *__restrict__ scal,
                                                 double *__restrict__ partial)
{
#pragma clang fp contract(off)      // products and sums round separately, as in xtv_epilogue
    constexpr int RS = 36;
    const int n16 = lane & 15, rg = lane >> 4;
    const int per_op = dm.per_op;
    double wgt[SLOTS];
    {
        unsigned long long w = 16;
        #pragma unroll
        for (int t = 0; t < SLOTS; ++t) { wgt[t] = t < dm.ndig ? (double)w : 0.0; w *= 49u; }
    }
    #pragma unroll
    for (int v = 0; v < NR; ++v) {
        const bool cut = HALF && v == NR - 1;      // the pass's last operand: only the residuals of its first 16 columns exist
        const int nsub = cut ? 16 / SLOTS : per_op;
        #pragma unroll
        for (int c = 0; c < CT; ++c) {
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int a = 0; a < 2; ++a)
                #pragma unroll
                for (int b = 0; b < (cut ? 1 : 2); ++b)      // D: column 16 b + lane % 16, SNP rows 16 a + 4 (lane / 16) + (0..3)
                    *reinterpret_cast<f32x4v *>(buf + (16 * b + n16) * RS + 16 * a + 4 * rg) =
                        f32x4v{acc[c][v][a][b][0], acc[c][v][a][b][1], acc[c][v][a][b][2], acc[c][v][a][b][3]};
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int o = lane + 64 * k, row = o & 31, sub = o >> 5;
                if (sub < nsub) {
                    const float *src = buf + sub * SLOTS * RS + row;
                    double x[SLOTS];
                    #pragma unroll
                    for (int t = 0; t < SLOTS; ++t) x[t] = (double)src[t * RS] * wgt[t];
                    double sum;
                    if ((SLOTS & (SLOTS - 1)) == 0) {
                        #pragma unroll
                        for (int off = SLOTS / 2; off > 0; off >>= 1)
                            #pragma unroll
                            for (int t = 0; t < off; ++t) x[t] = x[t] + x[t + off];
                        sum = x[0];
                    } else {
                        sum = 0.0;
                        #pragma unroll
                        for (int t = 0; t < SLOTS; ++t) sum += x[t];
                    }
                    const int rhs = v * per_op + sub;
                    if (cg0 + c < ncg)
                        partial[((int64_t)rhs * splits + split) * (ncg * 32) + (cg0 + c) * 32 + row] = sum * scal[4 * rhs + 1];
                }
            }
        }
    }
}

// Flat packing (DigitMode::flat, ten-digit format): residual j of the pass owns digit columns 10 j .. 10 j + 9 counted across the
// pass's operands, so a residual may begin in operand v and end in operand v + 1.  The operands' 32 x 32 tiles go through the
// wave's LDS buffer one after the other, as above; lane (row, q) takes the q-th residual that touches operand v, adds up ITS
// columns of this tile in digit order -- starting from 0 if the residual begins here, from the running sum it left in
// `carry[row]` in the previous operand otherwise -- and either stores the finished dot product or leaves the running sum for the
// next operand.  Products first, then one addition per digit from digit 0 upward: the operations, and their order, are those
// of xtv_epilogue16_s<.., 10, ..>, so the bits are the same wherever a residual sits.
template <int CT, int NR, int HALF>
__device__ __forceinline__ void xtv_epilogue16_flat(const f32x4a (&acc)[CT][NR][2][2], float *buf, int lane, int64_t cg0, int64_t ncg,
                                                    int split, int splits, DigitMode dm, const double *__restrict__ scal,
                                                    double *__restrict__ partial)
{
#pragma clang fp contract(off)      // products and sums round separately, as in xtv_epilogue
    constexpr int RS = 36, ND = 10;
    const int n16 = lane & 15, rg = lane >> 4;
    const int nres = dm.nres;
    double wgt[ND];
    {
        unsigned long long w = 16;
        #pragma unroll
        for (int t = 0; t < ND; ++t) { wgt[t] = (double)w; w *= 49u; }
    }
    double *carry = reinterpret_cast<double *>(buf + 32 * RS);       // [32 SNP rows], behind the tile
    #pragma unroll
    for (int c = 0; c < CT; ++c) {
        #pragma unroll
        for (int v = 0; v < NR; ++v) {
            const bool cut = HALF && v == NR - 1;      // the pass's last operand: only its first 16 columns exist
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int a = 0; a < 2; ++a)
                #pragma unroll
                for (int b = 0; b < (cut ? 1 : 2); ++b)
                    *reinterpret_cast<f32x4v *>(buf + (16 * b + n16) * RS + 16 * a + 4 * rg) =
                        f32x4v{acc[c][v][a][b][0], acc[c][v][a][b][1], acc[c][v][a][b][2], acc[c][v][a][b][3]};
            __builtin_amdgcn_wave_barrier();
            const int jlo = (32 * v) / ND, jhi = (32 * v + 31) / ND;       // residuals with a column in [32 v, 32 v + 32)
            #pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int o = lane + 64 * k, row = o & 31, j = jlo + (o >> 5);
                if (j <= jhi && j < nres) {
                    const int c0 = ND * j - 32 * v;                // column of digit 0 in this tile (negative: it began in the previous one)
                    const int tf = c0 < 0 ? -c0 : 0, tl = c0 + ND - 1 > 31 ? 31 - c0 : ND - 1;
                    double sum = tf == 0 ? 0.0 : carry[row];
                    #pragma unroll
                    for (int t = 0; t < ND; ++t)
                        if (t >= tf && t <= tl) {
                            const double x = (double)buf[(c0 + t) * RS + row] * wgt[t];
                            sum += x;
                        }
                    if (tl == ND - 1) {
                        if (cg0 + c < ncg)
                            partial[((int64_t)j * splits + split) * (ncg * 32) + (cg0 + c) * 32 + row] = sum * scal[4 * j + 1];
                    } else carry[row] = sum;           // (read in round k = 0 of the next operand; written here in a later round or after it)
                }
            }
        }
    }
}

// HALF = 1: the second 16-column fragment of the pass's LAST operand holds no residual (1 residual of 10 digits, or 2 of 8,
// in that operand: m = 3 j + 1 residuals in a pass) and its multiply-adds are left out -- 2 NR - 1 fragment items a step.
template <int NR, int CT, int WAVES, int D, int MODE = 0, int HALF = 0>       // MODE 3: timing probe, the odd 16-column fragments are skipped (result is NOT X'R); MODE 4: round 2's plain 8-byte LDS loads, which the compiler pairs into ds_read2_b64 (A/B for the bank-conflict fix; same result)
__global__ void __launch_bounds__(WAVES * 64, 1)
k_xtv_dma16(const uint4 *__restrict__ X, int64_t nbp, int64_t ncg, const uint4 *__restrict__ dig, const uint2 *__restrict__ dig2,
            int64_t dig_stride, int splits, DigitMode dm, const double *__restrict__ scal,
            double *__restrict__ partial /* [NR*per_op][splits][ncg*32] */)
{
    constexpr int S = D + 1;
    constexpr int DOS = WAVES * CT * 1024;
    constexpr int OPB = 3072;
    constexpr int DGT = NR * OPB;
    constexpr int STAGE = DOS + DGT;
    constexpr int NP = NR * 3;
    constexpr int PW = (NP + WAVES - 1) / WAVES;
    constexpr int L = CT + PW;
    constexpr int NI = 2 * NR - HALF;               // (operand, 16-column half) items of a step
    constexpr int ODD = NI & 1;                     // odd item count: the B double buffer alternates from step to step
    static_assert(S * STAGE <= 160 * 1024, "LDS ring too large");
    static_assert(D * L <= 63, "vmcnt range");
    static_assert(S * STAGE >= WAVES * (32 * 36 + 64) * 4, "the epilogue buffers overlay the ring");
    static_assert(HALF == 0 || MODE == 0, "probes run on full operands");
    __shared__ uint4 lds[S * STAGE / 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int split = blockIdx.x % splits;
    const int64_t grp = blockIdx.x / splits;
    const int64_t cg0 = (grp * WAVES + wave) * CT;
    const int64_t bps = (nbp + splits - 1) / splits;
    const int64_t b0 = split * bps;
    const int64_t b1 = (b0 + bps < nbp) ? b0 + bps : nbp;

    f32x4a acc[CT][NR][2][2];
    #pragma unroll
    for (int c = 0; c < CT; ++c)
        #pragma unroll
        for (int v = 0; v < NR; ++v)
            #pragma unroll
            for (int q = 0; q < 4; ++q)
                #pragma unroll
                for (int g = 0; g < 4; ++g) acc[c][v][q >> 1][q & 1][g] = 0.f;
    {
        const bool empty = b0 >= b1;               // see k_xtv_dma
        const int64_t bb0 = empty ? nbp - 1 : b0;
        const int nb = empty ? 1 : (int)(b1 - b0);
        const int amask = empty ? 0 : -1;
        const char *xs[CT];
        #pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int64_t cg = cg0 + c < ncg ? cg0 + c : ncg - 1;
            xs[c] = reinterpret_cast<const char *>(X) + (cg * nbp + bb0) * 1024;
        }
        const char *dsrc[PW]; int dstep[PW]; int doff[PW];
        #pragma unroll
        for (int u = 0; u < PW; ++u) {
            const int jj = wave + u * WAVES, j = jj < NP ? jj : NP - 1;
            const int op = j / 3, part = j - 3 * op;
            if (part < 2) {
                dsrc[u] = reinterpret_cast<const char *>(dig) + (op * dig_stride + (2 * bb0 + part) * 64) * 16;
                dstep[u] = 2048; doff[u] = DOS + op * OPB + part * 1024;
            } else {
                dsrc[u] = reinterpret_cast<const char *>(dig2) + (op * dig_stride + 2 * bb0 * 64) * 8;
                dstep[u] = 1024; doff[u] = DOS + op * OPB + 2048;
            }
        }
        const uint32_t voff = lane * 16;
        const uint32_t lds0 = (uint32_t)(uintptr_t)lds;           // low 32 bits of a generic LDS address = the byte offset M0 takes
        const uint32_t mydos = wave * CT * 1024;
        const char *ldsb = reinterpret_cast<const char *>(lds);
        // A fragment of SNP half a: the 8 bytes of row group (e, h) = (kq & 1, kq >> 1) of SNP 16 a + lane % 16
        const int kq = lane >> 4;
        const uint32_t aoff = (32 * (kq >> 1) + (lane & 15)) * 16 + 8 * (kq & 1);

        auto issue = [&](int ts, int st) {
            const int tb = ts < nb ? ts : nb - 1;
            const uint32_t base = lds0 + st * STAGE;
            #pragma unroll
            for (int c = 0; c < CT; ++c) glds16<true>(base + mydos + c * 1024, voff, xs[c] + (int64_t)tb * 1024);
            #pragma unroll
            for (int u = 0; u < PW; ++u) glds16<false>(base + doff[u], voff, dsrc[u] + (int64_t)tb * dstep[u]);
        };
        auto read_b = [&](int st, int item, i32x8 &b) {
            const int v = item >> 1, e = item & 1;
            const char *q = ldsb + st * STAGE + DOS + v * OPB;
            const i32x4v lo = *reinterpret_cast<const i32x4v *>(q + e * 1024 + lane * 16);
            const i32x2v hi = MODE == 4 ? *reinterpret_cast<const i32x2v *>(q + 2048 + e * 512 + lane * 8) : lds_read_b64(q + 2048 + e * 512 + lane * 8);
            b = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], 0, 0};
        };
        auto read_dos = [&](int st, i32x2v (&raw)[CT][2]) {
            #pragma unroll
            for (int c = 0; c < CT; ++c)
                #pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const char *q = ldsb + st * STAGE + mydos + c * 1024 + a * 256 + aoff;
                    raw[c][a] = MODE == 4 ? *reinterpret_cast<const i32x2v *>(q) : lds_read_b64(q);
                }
        };
        auto expand = [&](const i32x2v (&raw)[CT][2], i32x4v (&a)[CT][2]) {
            const int M = 0x33333333 & amask;
            #pragma unroll
            for (int c = 0; c < CT; ++c)
                #pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned x = raw[c][h][0], y = raw[c][h][1];
                    a[c][h] = i32x4v{(int)x & M, (int)(x >> 2) & M, (int)y & M, (int)(y >> 2) & M};
                }
        };

        #pragma unroll
        for (int s = 0; s < D; ++s) issue(s, s);
        i32x2v araw[CT][2];
        i32x4v A[2][CT][2];
        i32x8 B[2];
        wait_vm_barrier<(D - 1) * L>();
        read_dos(0, araw);
        read_b(0, 0, B[0]);
        expand(araw, A[0]);
        int st = 0;
#define MIH_DMA16_ITEM(P, I, BB)                                                                                   \
            _Pragma("unroll")                                                                                      \
            for (int c = 0; c < CT && !(MODE == 3 && ((I) & 1)); ++c) {                                            \
                acc[c][(I) >> 1][0][(I) & 1] = mfma16(A[P][c][0], BB, acc[c][(I) >> 1][0][(I) & 1]);              \
                acc[c][(I) >> 1][1][(I) & 1] = mfma16(A[P][c][1], BB, acc[c][(I) >> 1][1][(I) & 1]);              \
            }
#define MIH_DMA16_STEP(P, T)                                                                                       \
        {                                                                                                          \
            const int st_next = st + 1 == S ? 0 : st + 1, st_ld = st == 0 ? S - 1 : st - 1;                        \
            constexpr int PB = (P) * ODD;          /* item i of this step sits in B[(i + PB) & 1] */               \
            issue((T) + D, st_ld);                                                                                 \
            _Pragma("unroll")                                                                                      \
            for (int i = 0; i < NI - 1; ++i) {                                                                     \
                if (!(MODE == 3 && ((i + 1) & 1))) read_b(st, i + 1, B[(i + 1 + PB) & 1]);                         \
                MIH_DMA16_ITEM(P, i, B[(i + PB) & 1])                                                              \
                if (i == (NI > 2 ? NI / 2 - 1 : 0)) { wait_vm<D * L - CT>(); read_dos(st_next, araw); }            \
                if (i == (NI > 2 ? NI / 2 : 0)) expand(araw, A[(P) ^ 1]);                                          \
            }                                                                                                      \
            if (NI == 1) { wait_vm<D * L - CT>(); read_dos(st_next, araw); expand(araw, A[(P) ^ 1]); }             \
            wait_vm_barrier<(D - 1) * L>();                                                                        \
            read_b(st_next, 0, B[(NI + PB) & 1]);      /* = item 0 of the next step: ((P ^ 1) * ODD) & 1 */         \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            MIH_DMA16_ITEM(P, NI - 1, B[(NI - 1 + PB) & 1])                                                        \
            st = st_next;                                                                                          \
        }
        for (int t = 0; t < nb; t += 2) {
            MIH_DMA16_STEP(0, t)
            if (t + 1 < nb) MIH_DMA16_STEP(1, t + 1)
        }
#undef MIH_DMA16_STEP
#undef MIH_DMA16_ITEM
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (cg0 >= ncg) return;
    float *buf = reinterpret_cast<float *>(lds) + wave * (32 * 36 + 64);      // a 32 x 36 f32 tile + 32 running sums (flat packing)
    if (dm.flat) xtv_epilogue16_flat<CT, NR, HALF>(acc, buf, lane, cg0, ncg, split, splits, dm, scal, partial);
    else if (dm.slots == 10) xtv_epilogue16_s<CT, NR, 10, HALF>(acc, buf, lane, cg0, ncg, split, splits, dm, scal, partial);
    else xtv_epilogue16_s<CT, NR, 8, HALF>(acc, buf, lane, cg0, ncg, split, splits, dm, scal, partial);
}

// Combine slices, add the missing-entry correction, centre, scale -- for the `nres` residuals of a pass in one launch: partial,
// scal, r and out advance by one residual's stride each.  A thread keeps its column and walks the residuals (round 4; before,
// one grid row per residual read mu, sinv and the two missing-list bounds once per residual: 64 MB per residual at p = 1M and
// four slices, 40 MB now); the arithmetic of every (residual, column) is unchanged.
// (round 6) pl: the rows k_r_stats / k_res_peel took out of the fixed-point residual (peel.h) -- their terms g_ij r_i are added here in
// f64, ascending rows, behind the slices' sum; a missing genotype is stored as dosage 0 and gets its imputed value below, like the rest
__device__ __forceinline__ double xtv_finalize_col(int64_t j, const double *__restrict__ pu, int splits, int64_t pstride, double sum_r,
                                                   const double *__restrict__ ru, double m, double si, int64_t a, int64_t b,
                                                   const int32_t *__restrict__ miss_row, int center, int scale,
                                                   const double *__restrict__ pl, const uint32_t *__restrict__ X, int64_t nbp)
{
    // a NaN or +-Inf anywhere in the residual (its sum says so): the reference's floating-point mul! gives NaN or +-Inf in every column it
    // touches -- in every column of a centered matrix; the fixed point has no such value, so the answer is NaN in every column
    if (!(fabs(sum_r) <= 1.7976931348623157e308)) return __longlong_as_double(0x7ff8000000000000ll);
    double dot = 0.0;
    for (int s = 0; s < splits; ++s) dot += pu[(int64_t)s * pstride + j];
    if (pl) {
        const int np = (int)pl[0];
        for (int t = 0; t < np; ++t) {
            const int64_t i = (int64_t)pl[4 + t];
            const uint32_t g = (X[xword(nbp, j, i >> 4)] >> (2 * (int)(i & 15))) & 3u;
            dot += (double)g * pl[4 + kPeelMax + t];
        }
    }
    if (b > a) {
        double ms = 0.0;
        for (int64_t t = a; t < b; ++t) ms += ru[miss_row[t]];
        dot += m * ms;
    }
    if (center) dot -= m * sum_r;
    if (scale) dot *= si;
    return dot;
}
__global__ void __launch_bounds__(256)
k_xtv_finalize(const double *__restrict__ partial, int splits, int64_t pstride, int64_t p, int nres,
               const double *__restrict__ scal, const double *__restrict__ r, int64_t n,
               const double *__restrict__ mu, const double *__restrict__ sinv,
               const int64_t *__restrict__ miss_ptr, const int32_t *__restrict__ miss_row,
               int center, int scale, int impute, double *__restrict__ out,
               const int32_t *__restrict__ gate, int32_t gate_val, XtvSupportHook hook,
               const double *__restrict__ peel, const uint32_t *__restrict__ X, int64_t nbp)
{
    if (gate && *gate != gate_val) return;
    const int64_t pblocks = (p + 255) / 256;
    if ((int64_t)blockIdx.x >= pblocks) {          // the support of the current iterate (device-resident steps; nres == 1)
        const int c = *hook.cur;
        const int64_t t = ((int64_t)blockIdx.x - pblocks) * 256 + threadIdx.x;
        if (t >= *hook.cnt[c]) return;
        const int64_t j = hook.idx[c][t];
        const double m = mu[j], si = scale ? sinv[j] : 1.0;
        int64_t a = 0, b = 0;
        if (impute) { a = miss_ptr[j]; b = miss_ptr[j + 1]; }
        const double dot = xtv_finalize_col(j, partial, splits, pstride, scal[2], r, m, si, a, b, miss_row, center, scale, peel, X, nbp);
        const double av = si * dot;
        hook.gval[t] = dot; hook.A[t] = av; hook.B[t] = center ? -m * av : 0.0;
        return;
    }
    int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= p) return;
    const double m = mu[j], si = scale ? sinv[j] : 1.0;
    int64_t a = 0, b = 0;
    if (impute) { a = miss_ptr[j]; b = miss_ptr[j + 1]; }
    for (int u = 0; u < nres; ++u)
        out[(int64_t)u * p + j] = xtv_finalize_col(j, partial + (int64_t)u * splits * pstride, splits, pstride, scal[4 * u + 2],
                                                   r + (int64_t)u * n, m, si, a, b, miss_row, center, scale,
                                                   peel ? peel + (int64_t)u * kPeelStride : nullptr, X, nbp);
}

// ---- dense design matrix: out_j = sum_i D[i,j] r_i (one wave per column) ---------------
__device__ __forceinline__ double wave_sum(double v)
{
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
// One wave per column.  The matrix is read exactly once: nontemporal 16 B/lane loads, four of them in
// flight per lane (64 lanes x 64 B = 4 KB per wave-iteration), two accumulators per load slot; the
// fixed lane -> row mapping and the fixed final tree keep the sum bit-reproducible.
__global__ void __launch_bounds__(256)
k_xtv_dense(const double *__restrict__ D, int64_t n, int64_t p, const double *__restrict__ r,
            double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    int64_t j = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (j >= p) return;
    const double *col = D + j * n;
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if ((n & 1) == 0 && (((uintptr_t)col) & 15) == 0) {
        const f64x2 *cx = reinterpret_cast<const f64x2 *>(col);
        const f64x2 *rx = reinterpret_cast<const f64x2 *>(r);
        const int64_t n2 = n >> 1;
        int64_t i = lane;
        for (; i + 192 < n2; i += 256) {
            f64x2 x0 = __builtin_nontemporal_load(cx + i), x1 = __builtin_nontemporal_load(cx + i + 64);
            f64x2 x2 = __builtin_nontemporal_load(cx + i + 128), x3 = __builtin_nontemporal_load(cx + i + 192);
            f64x2 v0 = rx[i], v1 = rx[i + 64], v2 = rx[i + 128], v3 = rx[i + 192];
            a[0] = fma(x0.x, v0.x, a[0]); a[1] = fma(x0.y, v0.y, a[1]);
            a[2] = fma(x1.x, v1.x, a[2]); a[3] = fma(x1.y, v1.y, a[3]);
            a[4] = fma(x2.x, v2.x, a[4]); a[5] = fma(x2.y, v2.y, a[5]);
            a[6] = fma(x3.x, v3.x, a[6]); a[7] = fma(x3.y, v3.y, a[7]);
        }
        for (; i < n2; i += 64) {
            f64x2 x0 = __builtin_nontemporal_load(cx + i), v0 = rx[i];
            a[0] = fma(x0.x, v0.x, a[0]); a[1] = fma(x0.y, v0.y, a[1]);
        }
    } else {
        for (int64_t k = lane; k < n; k += 64) a[0] = fma(col[k], r[k], a[0]);
    }
    double s = wave_sum(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
    if (lane == 0) out[j] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Float32 storage of the dense matrix: the same loop with 16-B loads of four floats, products and sums in f64
__global__ void __launch_bounds__(256)
k_xtv_dense_f32(const float *__restrict__ D, int64_t n, int64_t p, const double *__restrict__ r, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    int64_t j = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (j >= p) return;
    const float *col = D + j * n;
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if ((n & 3) == 0 && (((uintptr_t)col) & 15) == 0) {
        const f32x4 *cx = reinterpret_cast<const f32x4 *>(col);
        const f64x2 *rx = reinterpret_cast<const f64x2 *>(r);
        const int64_t n4 = n >> 2;
        int64_t i = lane;
        for (; i + 64 < n4; i += 128) {
            f32x4 x0 = __builtin_nontemporal_load(cx + i), x1 = __builtin_nontemporal_load(cx + i + 64);
            f64x2 v0 = rx[2 * i], v1 = rx[2 * i + 1], v2 = rx[2 * (i + 64)], v3 = rx[2 * (i + 64) + 1];
            a[0] = fma((double)x0.x, v0.x, a[0]); a[1] = fma((double)x0.y, v0.y, a[1]);
            a[2] = fma((double)x0.z, v1.x, a[2]); a[3] = fma((double)x0.w, v1.y, a[3]);
            a[4] = fma((double)x1.x, v2.x, a[4]); a[5] = fma((double)x1.y, v2.y, a[5]);
            a[6] = fma((double)x1.z, v3.x, a[6]); a[7] = fma((double)x1.w, v3.y, a[7]);
        }
        for (; i < n4; i += 64) {
            f32x4 x0 = __builtin_nontemporal_load(cx + i);
            f64x2 v0 = rx[2 * i], v1 = rx[2 * i + 1];
            a[0] = fma((double)x0.x, v0.x, a[0]); a[1] = fma((double)x0.y, v0.y, a[1]);
            a[2] = fma((double)x0.z, v1.x, a[2]); a[3] = fma((double)x0.w, v1.y, a[3]);
        }
    } else {
        for (int64_t k = lane; k < n; k += 64) a[0] = fma((double)col[k], r[k], a[0]);
    }
    double s = wave_sum(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
    if (lane == 0) out[j] = s;
}

// The dense kernel of choice: the four waves of a block own four adjacent columns and walk them in steps of 256
// 16-B loads per column; the residual chunk of a step is staged ONCE per block in LDS (double buffered, one barrier
// per step) instead of being re-read from L2 by every wave, and the loads of step t+1 are issued before the FMAs of
// step t.  50 000 x 100 000 f64: 5.66 ms = 7.07 TB/s against 6.19 ms for k_xtv_dense (tools/dense_probe.hip).
// NRHS residual vectors (n apart in r, p apart in out) ride the same pass over D; per (column, residual) the
// arithmetic and its order are those of the NRHS = 1 kernel, so fused and single passes give the same bits.
// Fixed lane -> row mapping, fixed final tree: bit-reproducible.  Needs n % (16 / sizeof(T)) == 0 and a 16-B aligned D.
template <typename T, int NRHS>
__global__ void __launch_bounds__(256)
k_xtv_dense_lds(const T *__restrict__ D, int64_t n, int64_t p, const double *__restrict__ r, double *__restrict__ out)
{
    constexpr int E = 16 / (int)sizeof(T);         // matrix elements per 16-B load: 2 (f64) or 4 (f32)
    constexpr int RC = 128 * E;                    // f64x2 residual pairs per step (256 loads x E rows)
    constexpr int RK = RC / 256;                   // staged pairs per thread, step and residual
    typedef T vecT __attribute__((ext_vector_type(E)));
    __shared__ f64x2 rt[2][NRHS][RC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t jj = blockIdx.x * 4ll + wave, j = jj < p ? jj : p - 1;      // idle waves redo the last column
    const vecT *cx = reinterpret_cast<const vecT *>(D + j * n);
    const int64_t nv = n / E, nr2 = n >> 1, steps = (nv + 255) / 256;
    const vecT vzero = {};
    const f64x2 rzero = {0.0, 0.0};
    double a[NRHS][4][E];
    #pragma unroll
    for (int v = 0; v < NRHS; ++v)
        #pragma unroll
        for (int u = 0; u < 4; ++u)
            #pragma unroll
            for (int e = 0; e < E; ++e) a[v][u][e] = 0.0;
    #pragma unroll
    for (int v = 0; v < NRHS; ++v) {
        const f64x2 *rx = reinterpret_cast<const f64x2 *>(r + (int64_t)v * n);
        #pragma unroll
        for (int k = 0; k < RK; ++k) { const int t = threadIdx.x + 256 * k; rt[0][v][t] = t < nr2 ? rx[t] : rzero; }
    }
    __syncthreads();
    vecT x[4], xn[4];
    #pragma unroll
    for (int u = 0; u < 4; ++u) { const int64_t i = lane + 64 * u; x[u] = i < nv ? __builtin_nontemporal_load(cx + i) : vzero; }
    for (int64_t st = 0; st < steps; ++st) {
        const int buf = (int)(st & 1);
        f64x2 rn[NRHS][RK];
        #pragma unroll
        for (int v = 0; v < NRHS; ++v) {
            const f64x2 *rx = reinterpret_cast<const f64x2 *>(r + (int64_t)v * n);
            #pragma unroll
            for (int k = 0; k < RK; ++k) { const int64_t t = (st + 1) * RC + threadIdx.x + 256 * k; rn[v][k] = t < nr2 ? rx[t] : rzero; }
        }
        #pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t i = (st + 1) * 256 + lane + 64 * u; xn[u] = i < nv ? __builtin_nontemporal_load(cx + i) : vzero; }
        #pragma unroll
        for (int v = 0; v < NRHS; ++v) {
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                #pragma unroll
                for (int h = 0; h < E / 2; ++h) {
                    const f64x2 rv = rt[buf][v][(E / 2) * (lane + 64 * u) + h];
                    a[v][u][2 * h] = fma((double)x[u][2 * h], rv.x, a[v][u][2 * h]);
                    a[v][u][2 * h + 1] = fma((double)x[u][2 * h + 1], rv.y, a[v][u][2 * h + 1]);
                }
            }
        }
        #pragma unroll
        for (int v = 0; v < NRHS; ++v)
            #pragma unroll
            for (int k = 0; k < RK; ++k) rt[buf ^ 1][v][threadIdx.x + 256 * k] = rn[v][k];
        __syncthreads();
        #pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = xn[u];
    }
    #pragma unroll
    for (int v = 0; v < NRHS; ++v) {
        double su[4];
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            su[u] = a[v][u][0] + a[v][u][1];
            if (E == 4) su[u] = su[u] + (a[v][u][2] + a[v][u][3]);
        }
        const double sum = wave_sum((su[0] + su[1]) + (su[2] + su[3]));
        if (lane == 0 && jj < p) out[(int64_t)v * p + j] = sum;
    }
}

template <typename T>
static void launch_dense_lds(const T *D, const mih_mat *h, const double *r_dev, int m, double *out_dev, hipStream_t s)
{
    const dim3 grid((unsigned)((h->p + 3) / 4)), block(256);
    int v = 0;
    if constexpr (sizeof(T) == 8)          // eight f64 residual chunks fill the 64 KB of static LDS
        for (; v + 8 <= m; v += 8)
            hipLaunchKernelGGL((k_xtv_dense_lds<T, 8>), grid, block, 0, s, D, h->n, h->p, r_dev + (int64_t)v * h->n, out_dev + (int64_t)v * h->p);
    for (; v + 4 <= m; v += 4)
        hipLaunchKernelGGL((k_xtv_dense_lds<T, 4>), grid, block, 0, s, D, h->n, h->p, r_dev + (int64_t)v * h->n, out_dev + (int64_t)v * h->p);
    if (m - v >= 2) {
        hipLaunchKernelGGL((k_xtv_dense_lds<T, 2>), grid, block, 0, s, D, h->n, h->p, r_dev + (int64_t)v * h->n, out_dev + (int64_t)v * h->p);
        v += 2;
    }
    if (m - v == 1)
        hipLaunchKernelGGL((k_xtv_dense_lds<T, 1>), grid, block, 0, s, D, h->n, h->p, r_dev + (int64_t)v * h->n, out_dev + (int64_t)v * h->p);
}

constexpr int kMaxSplits = 16;
constexpr int kStatBlocks = 64;

// mih_fit_params::xtv_digits / the digits argument of mih_xtv_batched_fmt (ids = base * 100 + digits, include/mendeliht_hip.h)
static bool digit_mode(int id, DigitMode &dm)
{
    switch (id) {
    case -1:                                                  // auto (lock-step drivers): 4910, and 4908 for the residuals whose range allows it
    case 0:
    case 4910: dm = {49, 10, 3, 10, 53, 18}; return true;    // FP6, three residuals per operand, |R| < 2^54
    case 4908: dm = {49, 8, 4, 8, 42, 18}; return true;      // FP6, four per operand, |R| < 2^43 (opt-in fast mode)
    case 1316: dm = {13, 16, 2, 16, 56, 20}; return true;    // FP4, two per operand, |R| < 2^57
    case 1308: dm = {13, 8, 4, 8, 26, 20}; return true;      // FP4, four per operand, |R| < 2^27
    case 428:  dm = {4, 28, 1, 32, 53, 22}; return true;     // FP4, one per operand, |R| < 2^54
    }
    return false;
}
bool xtv_digits_valid(int digits) { DigitMode dm; return digit_mode(digits, dm); }

#ifdef MIH_PROBES
#include "xtv_probes.inc"       // launch-shape sweeps, round-1 kernel families, timing probes: measurement build only
#else
XtvTune xtv_tune(int digits) { XtvTune t; t.digits = digits; return t; }
static int dispatch_probe(const XtvTune &, int, bool, const mih_mat *, const uint4 *, const uint2 *, int64_t, int, DigitMode,
                          const double *, double *, hipStream_t, char *) { return -1; }
static int probe_splits(const XtvTune &) { return 0; }
#endif

// nr B operands per pass, each carrying per_op residuals.  The release library has ONE kernel per (format family, operand
// count):
//   FP6 planes (4910 / 4908):  k_xtv_dma16<nr,2,8,D,0,half>  -- the 16x16x128 LDS-DMA ring (12 residuals at n=500k, p=1M:
//                              28.5 ms against 34.3 ms for the 32x32x64 ring kernel and 40.0 ms register-staged)
//   FP4 planes, one operand:   k_xtv_dma<1,2,4,8,fp4>         -- the single-fit pass (17.6 ms = 88.8 % of the HBM peak)
//   FP4 planes, 2-4 operands:  k_xtv_mfma_lds<nr,..>          -- register-staged; only matrices too tall for FP6 slices
//                                                                (> 2^22 rows) or an explicit 1316 / 1308 / 428 get here
// `name` receives the dispatched kernel's name (profile records).
static int dispatch_xtv(const XtvTune &tn, int nr, bool half, const mih_mat *h, const uint4 *dig, const uint2 *dig2, int64_t dig_stride,
                        int splits, DigitMode dm, const double *scal, double *partial, hipStream_t s, char *name)
{
    {
        const int rc = dispatch_probe(tn, nr, half, h, dig, dig2, dig_stride, splits, dm, scal, partial, s, name);
        if (rc >= 0) return rc;
    }
#define MIH_DMA16(NRV, C, W, DD, HF) if (nr == NRV && (int)half == HF) { \
        int64_t groups = (h->ncg + W * C - 1) / (W * C); \
        hipLaunchKernelGGL((k_xtv_dma16<NRV, C, W, DD, 0, HF>), dim3((unsigned)(groups * splits)), dim3(W * 64), 0, s, \
                           reinterpret_cast<const uint4 *>(h->X), h->nbp, h->ncg, dig, dig2, dig_stride, splits, dm, scal, partial); \
        snprintf(name, 48, "k_xtv_dma16<%d,%d,%d,%d%s>", NRV, C, W, DD, HF ? ",half" : ""); \
        return MIH_OK; }
    if (dm.base == 49) {
        if (!dm.lay16) { set_error("FP6 digit planes need the 16-column layout"); return MIH_BAD_ARG; }
        // 5 operands = 15 residuals: 224 VGPRs, 155 KB of LDS at D = 4; 6 operands = 18 residuals: 254 VGPRs, 136 KB at D = 3
        // (ring depths 2, 3, 4 are equally fast once one step is ahead: 28.8 / 28.9 / 29.0 ms at 12 residuals)
        MIH_DMA16(6, 2, 8, 3, 1) MIH_DMA16(5, 2, 8, 4, 1) MIH_DMA16(4, 2, 8, 4, 1) MIH_DMA16(3, 2, 8, 4, 1) MIH_DMA16(2, 2, 8, 4, 1) MIH_DMA16(1, 2, 8, 4, 1)
        MIH_DMA16(6, 2, 8, 3, 0) MIH_DMA16(5, 2, 8, 4, 0) MIH_DMA16(4, 2, 8, 4, 0) MIH_DMA16(3, 2, 8, 4, 0) MIH_DMA16(2, 2, 8, 4, 0) MIH_DMA16(1, 2, 8, 4, 0)
        set_error("unsupported operand count %d", nr);
        return MIH_BAD_ARG;
    }
#undef MIH_DMA16
    if (nr == 1) {
        int64_t groups = (h->ncg + 4 * 2 - 1) / (4 * 2);
        hipLaunchKernelGGL((k_xtv_dma<1, 2, 4, 8, false, 0>), dim3((unsigned)(groups * splits)), dim3(4 * 64), 0, s,
                           reinterpret_cast<const uint4 *>(h->X), h->nbp, h->ncg, dig, dig2, dig_stride, splits, dm, scal, partial);
        snprintf(name, 48, "k_xtv_dma<1,2,4,8,fp4>");
        return MIH_OK;
    }
#define MIH_LDS(NRV, C, RB, W) if (nr == NRV) { \
        int64_t groups = (h->ncg + W * C - 1) / (W * C); \
        hipLaunchKernelGGL((k_xtv_mfma_lds<NRV, C, RB, 0, W>), dim3((unsigned)(groups * splits)), dim3(W * 64), 0, s, \
                           reinterpret_cast<const uint4 *>(h->X), h->nbp, h->ncg, dig, dig2, dig_stride, splits, dm, scal, partial); \
        snprintf(name, 48, "k_xtv_mfma_lds<%d,%d,%d,%d,fp4>", NRV, C, RB, W); \
        return MIH_OK; }
    MIH_LDS(4, 2, 2, 8) MIH_LDS(3, 2, 2, 8) MIH_LDS(2, 4, 1, 4)
#undef MIH_LDS
    set_error("unsupported operand count %d", nr);
    return MIH_BAD_ARG;
}

// Row slices of the library-default kernels.  A slice should hold about 50 000 rows or more -- every (column group,
// slice) work item pays a prologue and a 16-accumulator epilogue, and short items lose to that: at n = 50 000 one
// slice runs at 87 % of the HBM peak and eight at 75 %, at n = 10 000 it is 74 % against 38 % -- but there must be
// enough workgroups to fill 256 CUs.  At n = 500 000 round 1's kernels were best with eight slices (87.0 % against 83.3 % for
// one); with the LDS-DMA ring kernels FOUR are (round 3, tools/sweep_slices*.py, alternated in one process: single-fit pass
// 17.57 ms against 17.63 ms with eight, 18 residuals 38.9-39.1 against 39.2-39.4 ms) -- and the count must divide the 8 XCDs
// (slice = blockIdx % slices; five slices: 18.76 ms).
static int auto_splits(const mih_mat *h, const XtvTune &tn)
{
    if (tn.slices >= 1 && tn.slices <= kMaxSplits) return tn.slices;       // measurement build only
    int s = 1;
    while (s < 4 && h->n >= 100000ll * s) s *= 2;
    const int64_t groups = (h->ncg + 15) / 16;                 // workgroups per slice of the widest launch shape
    while (s < 16 && groups * s < 2048 && h->nbp / (2 * s) >= 8) s *= 2;
    return s;
}

// FP6 digit planes are written in the 16-column layout when the pass runs on the 16x16x128 kernels (always, in the release
// library; the measurement build also has 32x32x64 shapes)
static bool xtv_lay16(const DigitMode &dm, const XtvTune &tn)
{
    return dm.base == 49 && tn.variant < 0 && (tn.multi_variant == 0 || (tn.multi_variant >= 40 && tn.multi_variant < 50));
}

// Flat packing of the digit columns (DigitMode::flat): the ten-digit format on the product's 16x16x128 kernels.  Three residuals
// per operand leave 2 of 32 columns idle; packed back to back, 19 residuals fit the 192 columns of a six-operand pass instead of
// 18, 16 fit five operands instead of five and a half.  (The other launch shapes of the measurement build keep the per-operand
// layout: same bits, they are the cross-check.)
static bool xtv_flat(const DigitMode &dm, const XtvTune &tn)
{
    return xtv_lay16(dm, tn) && tn.multi_variant == 0 && dm.slots * dm.per_op < 32 && tn.max_nr >= 4;
}
// residuals a full pass of the fused kernel scores
static int xtv_pass_residuals(const DigitMode &dm, const XtvTune &tn)
{
    return xtv_flat(dm, tn) ? (tn.max_ops * 32) / dm.slots : tn.max_ops * dm.per_op;
}

static void choose_mode(const mih_mat *h, const XtvTune &tn, bool batched, DigitMode &dm)
{
    digit_mode(tn.digits, dm);
    if (tn.digits == 0 || tn.digits == -1) {
        // Library default.  The workspace of a single fit scores one residual per pass: 28 sparse base-4 digit
        // columns make the fastest single-operand pass (17.86 ms against 17.99 ms for 16 base-13 columns and
        // 18.4 ms for FP6 planes at n=500k, p=1M; tools/sweep_fp6_single.py).  Every fused multi-RHS context
        // (cv_iht, model paths, multivariate fits, init_beta, mih_xtv_batched) uses the FP6 format whatever the
        // number of residuals in a call, so results never depend on how residuals are batched.  Larger digits mean
        // shorter exact row slices (2^18 rows in base 49, 2^20 in base 13, 2^22 in base 4), so very tall matrices
        // step down.
        if (!batched) digit_mode(428, dm);
        else if (tn.variant >= 0 || h->n_pad > ((int64_t)kMaxSplits << dm.rows_log2)) digit_mode(1316, dm);
        if (h->n_pad > ((int64_t)kMaxSplits << dm.rows_log2)) digit_mode(428, dm);
    }
}

int xtv_lockstep_width(const mih_mat *h, const XtvTune &tn)
{
    if (h->kind != 0) return 16;
    DigitMode dm;
    choose_mode(h, tn, true, dm);
    if (xtv_lay16(dm, tn)) return 2 * xtv_pass_residuals(dm, tn);    // two lanes of one full pass each
    return 8 * dm.per_op;
}

int xtv_work_init(const mih_mat *h, XtvWork &w, int m, const XtvTune &tune, bool batched)
{
    w.tune = tune;
    if (h->kind != 0) return MIH_OK;
    if (!xtv_digits_valid(tune.digits)) { set_error("residual format must be 0 (default), -1 (auto in lock-step drivers), 4910, 4908, 1316, 1308 or 428"); return MIH_BAD_ARG; }
    int64_t nblk = h->nbp * 2;
    choose_mode(h, tune, batched, w.dm);
    w.has_alt = tune.digits == -1 && batched && w.dm.base == 49 && w.dm.ndig == 10;      // (a matrix too tall for FP6 slices stepped down: no 43-bit twin)
    w.use_alt = false;
    if (w.dm.base == 49 && tune.variant >= 0) {         // only an explicitly requested FP6 format can get here (measurement build)
        set_error("the FP6 residual formats need the default kernel (variant -1)");
        return MIH_BAD_ARG;
    }
    int ops = (m + w.dm.per_op - 1) / w.dm.per_op;
    size_t rhs_cap = (size_t)ops * w.dm.per_op;
    if (xtv_flat(w.dm, tune)) {          // flat packing: ceil(10 m' / 32) operands plus at most one partly used per pass, for any m' <= m
        const int cap = xtv_pass_residuals(w.dm, tune), npass = (m + cap - 1) / cap;
        if (npass <= kMaxFlatPasses) { ops = (m * w.dm.slots + 31) / 32 + npass; rhs_cap = (size_t)m; }
    }
    w.ops_cap = ops; w.rhs_cap = rhs_cap;
    const size_t lanes = (size_t)w.ops_cap * (size_t)nblk * 64;
    const size_t dwords = lanes * (w.dm.base == 49 ? 6 : 4);
    MIH_TRY(w.digits.alloc(dwords));
    MIH_TRY(w.partial.alloc((size_t)kMaxSplits * rhs_cap * (size_t)h->ncg * 32));
    MIH_TRY(w.scal.alloc(rhs_cap * 4 + (size_t)m * kStatBlocks * 2));
    MIH_TRY(w.stat_done.alloc(rhs_cap));
    MIH_TRY(w.peel.alloc(rhs_cap * kPeelStride));
    MIH_HIP(hipMemsetAsync(w.peel.p, 0, sizeof(double) * rhs_cap * kPeelStride, h->stream));       // (no rows peeled; the guards' running counts at zero)
    MIH_HIP(hipMemset(w.stat_done.p, 0, sizeof(unsigned) * rhs_cap));       // k_r_stats leaves its counters at zero
    MIH_HIP(hipMemsetAsync(w.digits.p, 0, sizeof(*w.digits.p) * dwords, h->stream));
    MIH_HIP(hipMemsetAsync(w.scal.p, 0, sizeof(double) * rhs_cap * 4, h->stream));
    MIH_HIP(hipStreamSynchronize(h->stream));
    w.m_cap = m; w.splits_cap = kMaxSplits;
    return MIH_OK;
}

// a lane's pass takes its turn behind the pass queued last on this matrix's lanes (PassOrder, common.h)
struct PassTurn {
    XtvWork &w; hipStream_t s; bool held = false;
    PassTurn(XtvWork &ww, hipStream_t ss) : w(ww), s(ss)
    {
        if (!w.order || !w.pass_done) return;
        w.order->mu.lock(); held = true;
        if (w.order->last && w.order->last != w.pass_done) (void)hipStreamWaitEvent(s, w.order->last, 0);
    }
    ~PassTurn()
    {
        if (!held) return;
        (void)hipEventRecord(w.pass_done, s);
        w.order->last = w.pass_done;
        w.order->mu.unlock();
    }
};

// measurement hook: how often the outlier guard (peel.h) of this workspace's residual slots has fired since the last call
// (MIH_CNT_PEELED_RESIDUALS); synchronises the stream
void xtv_count_peels(const mih_mat *h, XtvWork &w, hipStream_t s)
{
    if (!h->prof->on || !w.peel.p || w.rhs_cap == 0) return;
    // (the whole buffer in ONE plain copy, <= 20 KB: a strided hipMemcpy2DAsync of the counts alone took ~6 ms per call in a process
    // that also holds PyTorch's HIP runtime -- bench.py's host-driven A/B read 0.64 ms outside the pass instead of 0.31)
    std::vector<double> c(w.rhs_cap * kPeelStride, 0.0);
    if (hipMemcpyAsync(c.data(), w.peel.p, sizeof(double) * c.size(), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return; }
    double total = 0.0;
    for (size_t v = 0; v < w.rhs_cap; ++v) total += c[v * kPeelStride + 3];
    h->prof->count(MIH_CNT_PEELED_RESIDUALS, (int64_t)(total - w.peels_counted));
    w.peels_counted = total;
}

int xtv_device(const mih_mat *h, XtvWork &w, const double *r_dev, int m, double *out_dev, hipStream_t s)
{
    const XtvTune &tn = w.tune;
    if (h->kind == 1) {
        const bool lds_ok = tn.variant < 0 && (((uintptr_t)(h->Df ? (const void *)h->Df : (const void *)h->D)) & 15) == 0
                            && h->n % (h->Df ? 4 : 2) == 0;
        PassRecord rec;
        const bool prof = prof_begin(h, s, rec);
        if (lds_ok) {
            if (h->Df) launch_dense_lds<float>(h->Df, h, r_dev, m, out_dev, s);
            else launch_dense_lds<double>(h->D, h, r_dev, m, out_dev, s);
        } else {
            for (int v = 0; v < m; ++v) {
                if (h->Df) hipLaunchKernelGGL(k_xtv_dense_f32, dim3((unsigned)((h->p + 3) / 4)), dim3(256), 0, s, h->Df, h->n, h->p,
                                              r_dev + (int64_t)v * h->n, out_dev + (int64_t)v * h->p);
                else hipLaunchKernelGGL(k_xtv_dense, dim3((unsigned)((h->p + 3) / 4)), dim3(256), 0, s, h->D, h->n, h->p,
                                        r_dev + (int64_t)v * h->n, out_dev + (int64_t)v * h->p);
            }
        }
        if (prof) {
            rec.residuals = m; rec.operands = m; rec.stream_tag = w.stream_tag;
            snprintf(rec.kernel, sizeof(rec.kernel), "%s<%s>", lds_ok ? "k_xtv_dense_lds" : "k_xtv_dense", h->Df ? "f32" : "f64");
            prof_end(h, s, rec);
        }
        MIH_HIP(hipGetLastError());
        return MIH_OK;
    }
    if (m > w.m_cap) { set_error("X'r workspace too small"); return MIH_BAD_ARG; }
    int splits = probe_splits(tn);
    if (splits <= 0) splits = auto_splits(h, tn);
    // exactness of the f32 accumulators: |g/2 * d/2| <= 1 (base 4) or 4 (base 13) in units of 1/4, so a row
    // slice may hold at most 2^22 / 2^20 rows
    DigitMode dm = (w.use_alt && w.has_alt) ? w.dm_alt : w.dm;
    dm.lay16 = xtv_lay16(dm, tn);
    const int64_t need = (h->n_pad + (1ll << dm.rows_log2) - 1) >> dm.rows_log2;
    if (need > w.splits_cap) { set_error("n = %lld rows needs more than %d row slices for exact accumulation", (long long)h->n, w.splits_cap); return MIH_BAD_DIM; }
    if (splits < need) splits = (int)need;
    if (splits > h->nbp) splits = (int)h->nbp;
    if (splits > w.splits_cap) splits = w.splits_cap;
    const int64_t nblk = h->nbp * 2, pstride = h->ncg * 32;
    const int per_op = dm.per_op;
    const int nops = (m + per_op - 1) / per_op;
    double *part = w.scal.p + w.rhs_cap * 4;
    dm.gate = w.gate; dm.gate_val = w.gate_val;
    if (!w.stats_done)
        hipLaunchKernelGGL(k_r_stats, dim3(kStatBlocks, (unsigned)m), dim3(256), 0, s, r_dev, h->n, m, part, w.stat_done.p, dm.ebits, w.scal.p, w.gate, w.gate_val, w.peel.p);
    uint4 *dig_all = reinterpret_cast<uint4 *>(w.digits.p);
    uint2 *dig2_all = reinterpret_cast<uint2 *>(w.digits.p + (size_t)w.ops_cap * (size_t)nblk * 64 * 4);   // FP6 only
    FlatPasses fp{};
    if (xtv_flat(dm, tn)) {
        // the residuals are dealt out evenly over as few passes as hold them; pass q scores residuals u0[q] .. u0[q+1] from the
        // operands t0[q] .. t0[q+1] (ceil(10 cnt / 32) of them: never more than the per-operand layout needs)
        const int cap = xtv_pass_residuals(dm, tn), npass = (m + cap - 1) / cap;
        if (npass <= kMaxFlatPasses) {
            dm.flat = 1;
            fp.npass = npass;
            int u = 0, t = 0;
            for (int q = 0; q < npass; ++q) {
                const int cnt = m / npass + (q < m % npass ? 1 : 0);
                fp.u0[q] = u; fp.t0[q] = t;
                u += cnt; t += (cnt * dm.slots + 31) / 32;
            }
            fp.u0[npass] = u; fp.t0[npass] = t;
        }
    }
    // INVARIANT of the flat packing: k_digits writes only the digit columns of THIS call's m residuals; the unused tail
    // columns of a pass's last operand keep whatever an earlier call with another m left there.  That is harmless because (1) the
    // columns of the MFMA's B operand are independent -- a stale column only produces accumulator columns nobody reads -- and
    // (2) xtv_epilogue16_flat stops at j >= nres.  The lock-step lanes exercise exactly this in every cross-validation (one workspace,
    // a falling residual count from round to round, every loss checked by the tests of the full configs[3] grid),
    // and test_flat_packing_ignores_what_an_earlier_pass_left_in_the_tail_columns runs 19 residuals and then 1 .. 18 on ONE workspace.
    hipLaunchKernelGGL(k_digits, dim3((unsigned)((nblk + 3) / 4), (unsigned)(dm.flat ? m : nops * per_op)), dim3(256), 0, s, r_dev, h->n, nblk, m, dm,
                       w.scal.p, dig_all, dig2_all, fp, (w.stats_done && m == 1) ? w.shook : XtvStatsHook(), w.peel.p);
    for (int q = 0; dm.flat && q < fp.npass; ++q) {
        const int u0 = fp.u0[q], u1 = fp.u0[q + 1], t = fp.t0[q], nr = fp.t0[q + 1] - t;
        DigitMode dq = dm;
        dq.nres = u1 - u0;
        double *partial = w.partial.p + (int64_t)u0 * splits * pstride;
        PassRecord rec;
        const bool prof = prof_begin(h, s, rec);
        const bool half = tn.half && (u1 - u0) * dm.slots - 32 * (nr - 1) <= 16;      // the last operand's second fragment holds no column
        char name[48] = {0};
        int rc;
        {
            PassTurn turn(w, s);
            rc = dispatch_xtv(tn, nr, half, h, dig_all + (int64_t)t * nblk * 64, dig2_all + (int64_t)t * nblk * 64, nblk * 64, splits, dq,
                              w.scal.p + 4 * u0, partial, s, name);
        }
        if (prof) {
            rec.residuals = u1 - u0; rec.operands = nr; rec.stream_tag = w.stream_tag;
            memcpy(rec.kernel, name, sizeof(rec.kernel));
            prof_end(h, s, rec);
        }
        if (rc) return rc;
        hipLaunchKernelGGL(k_xtv_finalize, dim3((unsigned)((h->p + 255) / 256)), dim3(256), 0, s,
                           partial, splits, pstride, h->p, u1 - u0, w.scal.p + 4 * u0, r_dev + (int64_t)u0 * h->n, h->n, h->mu, h->sinv,
                           h->miss_ptr, h->miss_row, h->center, h->scale, h->impute && h->total_missing > 0, out_dev + (int64_t)u0 * h->p, w.gate, w.gate_val, XtvSupportHook(),
                           w.peel.p + (int64_t)u0 * kPeelStride, h->X, h->nbp);
    }
    if (dm.flat) { MIH_HIP(hipGetLastError()); return MIH_OK; }
    for (int t = 0; t < nops;) {          // t counts B operands
        int nr = (nops - t >= 4 && nops - t != 5 && tn.max_nr >= 4) ? 4 : ((nops - t == 3 || nops - t == 5) && tn.max_nr >= 4) ? 3
                 : (nops - t >= 2 && tn.max_nr >= 2) ? 2 : 1;
        if (dm.lay16 && tn.max_nr >= 4) {          // up to max_ops operands a pass, the remainder split evenly
            const int rem = nops - t, mo = tn.max_ops, passes = (rem + mo - 1) / mo;
            nr = (rem + passes - 1) / passes;
        }
        const int u0 = t * per_op;        // first residual of this pass
        double *partial = w.partial.p + (int64_t)u0 * splits * pstride;
        const uint4 *dig = dig_all + (int64_t)t * nblk * 64;
        const uint2 *dig2 = dig2_all + (int64_t)t * nblk * 64;
        PassRecord rec;
        const bool prof = prof_begin(h, s, rec);
        const int u1 = std::min((t + nr) * per_op, m);
        // residuals in the pass's last operand; if they end within its first 16 columns the second fragment is left out
        const bool half = dm.lay16 && tn.half && (u1 - (t + nr - 1) * per_op) * dm.slots <= 16;
        char name[48] = {0};
        int rc;
        {
            PassTurn turn(w, s);
            rc = dispatch_xtv(tn, nr, half, h, dig, dig2, nblk * 64, splits, dm, w.scal.p + 4 * u0, partial, s, name);
        }
        if (prof) {
            rec.residuals = u1 - u0; rec.operands = nr; rec.stream_tag = w.stream_tag;
            memcpy(rec.kernel, name, sizeof(rec.kernel));
            prof_end(h, s, rec);
        }
        if (rc) return rc;
        if (u1 > u0) {
            const bool hooked = w.hook.cur != nullptr && m == 1;
            hipLaunchKernelGGL(k_xtv_finalize, dim3((unsigned)((h->p + 255) / 256 + (hooked ? w.hook.blocks : 0))), dim3(256), 0, s,
                               w.partial.p + (int64_t)u0 * splits * pstride, splits, pstride, h->p, u1 - u0,
                               w.scal.p + 4 * u0, r_dev + (int64_t)u0 * h->n, h->n, h->mu, h->sinv, h->miss_ptr, h->miss_row,
                               h->center, h->scale, h->impute && h->total_missing > 0, out_dev + (int64_t)u0 * h->p, w.gate, w.gate_val,
                               hooked ? w.hook : XtvSupportHook(), w.peel.p + (int64_t)u0 * kPeelStride, h->X, h->nbp);
        }
        t += nr;
    }
    MIH_HIP(hipGetLastError());
    return MIH_OK;
}

}  // namespace mih

using namespace mih;

extern "C" {

int mih_profile_enable(const mih_mat *h, int on)
{
    if (!h) { set_error("null matrix handle"); return MIH_BAD_ARG; }
    Profile &pf = *h->prof;
    std::lock_guard<std::mutex> g(pf.mu);
    if (on && !pf.on) {
        MIH_HIP(hipSetDevice(h->device));
        if (pf.origin) { (void)hipEventDestroy(pf.origin); pf.origin = nullptr; }
        MIH_HIP(hipEventCreate(&pf.origin));
        MIH_HIP(hipEventRecord(pf.origin, h->stream));
    }
    pf.on = on != 0;
    return MIH_OK;
}

int mih_profile_read(const mih_mat *h, double *xtv_kernel_ms, int64_t *xtv_launches, int reset)
{
    if (!h) { set_error("null matrix handle"); return MIH_BAD_ARG; }
    Profile &pf = *h->prof;
    std::lock_guard<std::mutex> g(pf.mu);
    pf.drain();
    double ms = 0.0;
    for (const auto &r : pf.done) ms += r.ms;
    if (xtv_kernel_ms) *xtv_kernel_ms = ms;
    if (xtv_launches) *xtv_launches = (int64_t)pf.done.size();
    if (reset) pf.done.clear();
    return MIH_OK;
}

int mih_profile_passes(const mih_mat *h, mih_pass_record *out, int64_t cap, int64_t *n, int reset)
{
    if (!h || !n) { set_error("null argument"); return MIH_BAD_ARG; }
    Profile &pf = *h->prof;
    std::lock_guard<std::mutex> g(pf.mu);
    pf.drain();
    *n = (int64_t)pf.done.size();
    if (out) for (int64_t i = 0; i < cap && i < *n; ++i) out[i] = pf.done[(size_t)i];
    if (reset) pf.done.clear();
    return MIH_OK;
}

int mih_profile_exchange(const mih_mat *h, double *ms4, int64_t *count4, int reset)
{
    if (!h || !ms4 || !count4) { set_error("null argument"); return MIH_BAD_ARG; }
    Profile &pf = *h->prof;
    std::lock_guard<std::mutex> g(pf.mu);
    for (auto &r : pf.xopen) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { pf.xms[r.kind] += ms; ++pf.xcount[r.kind]; }
        else (void)hipGetLastError();
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    pf.xopen.clear();
    for (int i = 0; i < 4; ++i) { ms4[i] = pf.xms[i]; count4[i] = pf.xcount[i]; if (reset) { pf.xms[i] = 0.0; pf.xcount[i] = 0; } }
    return MIH_OK;
}

int mih_profile_counters(const mih_mat *h, int64_t *out, int reset)
{
    if (!h || !out) { set_error("null argument"); return MIH_BAD_ARG; }
    Profile &pf = *h->prof;
    std::lock_guard<std::mutex> g(pf.mu);
    for (int i = 0; i < MIH_PROFILE_NCOUNTERS; ++i) { out[i] = pf.counters[i]; if (reset) pf.counters[i] = 0; }
    return MIH_OK;
}

int mih_xtv_algorithmic_bytes(const mih_mat *h, int m, double *bytes)
{
    if (!h || !bytes || m < 1) return MIH_BAD_ARG;
    if (h->kind == 0)
        *bytes = (double)h->p * (double)((h->n + 3) / 4) + 8.0 * m * ((double)h->n + (double)h->p) + 16.0 * (double)h->p;
    else
        *bytes = (h->Df ? 4.0 : 8.0) * (double)h->n * (double)h->p + 8.0 * m * ((double)h->n + (double)h->p);
    return MIH_OK;
}

int mih_xtv_batched_fmt(const mih_mat *h, const double *R, int m, int digits, double *OUT)
{
    if (!h || !R || !OUT || m < 1) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(h->device));
    XtvWork w;
    MIH_TRY(xtv_work_init(h, w, m, xtv_tune(digits)));
    DevBuf<double> r, out;
    MIH_TRY(r.alloc((size_t)m * h->n));
    MIH_TRY(out.alloc((size_t)m * h->p));
    MIH_HIP(hipMemcpyAsync(r.p, R, sizeof(double) * (size_t)m * h->n, hipMemcpyHostToDevice, h->stream));
    MIH_TRY(xtv_device(h, w, r.p, m, out.p, h->stream));
    MIH_HIP(hipMemcpyAsync(OUT, out.p, sizeof(double) * (size_t)m * h->p, hipMemcpyDeviceToHost, h->stream));
    MIH_HIP(hipStreamSynchronize(h->stream));
    xtv_count_peels(h, w, h->stream);
    return MIH_OK;
}

int mih_xtv_batched(const mih_mat *h, const double *R, int m, double *OUT)
{
    return mih_xtv_batched_fmt(h, R, m, 0, OUT);
}

int mih_xtv(const mih_mat *h, const double *r, double *out)
{
    return mih_xtv_batched_fmt(h, r, 1, 0, out);
}

#ifdef MIH_PROBES
// include/mendeliht_hip_probes.h: a sequence of residual counts on ONE fused-pass workspace (what a lock-step lane does)
int mih_probe_xtv_sequence(const mih_mat *h, const double *R, int mcap, const int *ms, int nms, int digits, double *OUT)
{
    if (!h || !R || !ms || !OUT || mcap < 1 || nms < 1) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    for (int i = 0; i < nms; ++i) if (ms[i] < 1 || ms[i] > mcap) { set_error("ms[%d] = %d outside 1..%d", i, ms[i], mcap); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(h->device));
    XtvWork w;
    MIH_TRY(xtv_work_init(h, w, mcap, xtv_tune(digits)));
    DevBuf<double> r, out;
    MIH_TRY(r.alloc((size_t)mcap * h->n));
    MIH_TRY(out.alloc((size_t)mcap * h->p));
    MIH_HIP(hipMemcpyAsync(r.p, R, sizeof(double) * (size_t)mcap * h->n, hipMemcpyHostToDevice, h->stream));
    size_t off = 0;
    for (int i = 0; i < nms; ++i) {
        MIH_TRY(xtv_device(h, w, r.p, ms[i], out.p, h->stream));
        MIH_HIP(hipMemcpyAsync(OUT + off, out.p, sizeof(double) * (size_t)ms[i] * h->p, hipMemcpyDeviceToHost, h->stream));
        MIH_HIP(hipStreamSynchronize(h->stream));
        off += (size_t)ms[i] * h->p;
    }
    return MIH_OK;
}
#endif

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

int mih_bench_xtv(const mih_mat *h, int digits, int m, int iters, int warmup, uint64_t seed, float *ms_per_pass, double *checksum)
{
    if (!h || iters < 1 || !ms_per_pass || m < 1) return MIH_BAD_ARG;
    MIH_HIP(hipSetDevice(h->device));
    XtvWork w;
    MIH_TRY(xtv_work_init(h, w, m, xtv_tune(digits), m > 1));       // m = 1: the single-fit workspace, like IhtVar
    DevBuf<double> r, out, cs;
    MIH_TRY(r.alloc((size_t)h->n * m));
    MIH_TRY(out.alloc((size_t)h->p * m));
    MIH_TRY(cs.alloc(1));
    hipStream_t s = h->stream;
    int rc = MIH_OK;
    hipLaunchKernelGGL(k_fill_random, dim3((unsigned)((h->n * m + 255) / 256)), dim3(256), 0, s, r.p, h->n * m, seed);
    for (int i = 0; i < warmup && !rc; ++i) rc = xtv_device(h, w, r.p, m, out.p, s);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters && !rc; ++i) rc = xtv_device(h, w, r.p, m, out.p, s);
    (void)hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(e, "bench sync", __FILE__, __LINE__);
    *ms_per_pass = ms / iters;
    if (checksum) {
        hipLaunchKernelGGL(k_checksum, dim3(1), dim3(256), 0, s, out.p, h->p * m, cs.p);
        MIH_HIP(hipMemcpyAsync(checksum, cs.p, sizeof(double), hipMemcpyDeviceToHost, s));
        MIH_HIP(hipStreamSynchronize(s));
    }
    return MIH_OK;
}

}  // extern "C"
