// snp.hip -- device-resident design matrices: upload/transcode of PLINK 2-bit
// columns, column statistics (mu, sinv), missing-entry side lists, synthetic
// generators, export.  Replaces the SnpLinAlg constructor the reference calls at
// src/wrapper.jl:68-69 (SnpArrays.jl linalg_direct.jl; semantics SURVEY.md 8c).
#include "common.h"
#include <memory>
#include <cstdarg>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>

namespace mih {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? MIH_OOM : MIH_HIP_ERROR;
}

// PLINK codes -> dosage codes for 16 packed genotypes: 00->00, 01->00 (missing), 10->01, 11->10.
__device__ __forceinline__ uint32_t plink_to_dosage(uint32_t w)
{
    uint32_t hi = w & 0xAAAAAAAAu;
    uint32_t both = (w << 1) & hi;
    return (hi >> 1) + (both >> 1);
}
__device__ __forceinline__ uint32_t plink_missing_mask(uint32_t w)   // bit 2s set where code == 01
{
    return (~w >> 1) & w & 0x55555555u;
}

__device__ __forceinline__ uint32_t lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Per-column counters of one workgroup (8 threads share a column) -> 96 integer atomics per block.
__device__ __forceinline__ void flush_counts(int32_t c1, int32_t c2, int32_t cm, int64_t cg, int64_t p,
                                             int32_t *__restrict__ cnt)
{
    __shared__ int32_t red[3][32];
    if (threadIdx.x < 96) red[threadIdx.x / 32][threadIdx.x % 32] = 0;
    __syncthreads();
    int m = threadIdx.x & 31;
    atomicAdd(&red[0][m], c1); atomicAdd(&red[1][m], c2); atomicAdd(&red[2][m], cm);
    __syncthreads();
    if (threadIdx.x < 96) {
        int k = threadIdx.x / 32, mm = threadIdx.x % 32;
        int64_t j = cg * 32 + mm;
        if (j < p && red[k][mm]) atomicAdd(&cnt[3 * j + k], red[k][mm]);
    }
}

// grid (chunks of block pairs, column groups).  Thread (w, lane) builds the lane records of block
// pairs w, w+4, ... of its chunk: transcode raw PLINK bytes to dosage codes and count.
constexpr int kBpPerBlock = 64;
__global__ void __launch_bounds__(256)
k_transcode(const uint8_t *__restrict__ raw, int64_t raw_stride, int64_t n, int64_t p, int64_t col0,
            int64_t ncols, uint4 *__restrict__ X, int64_t nbp, int32_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    const int64_t cg = col0 / 32 + blockIdx.y;
    const int64_t j = cg * 32 + m, jl = j - col0;
    const bool live = j < p && jl < ncols;
    const uint8_t *src = raw + (live ? jl : 0) * raw_stride;
    const int64_t nbytes = (n + 3) >> 2;
    int32_t c1 = 0, c2 = 0, cm = 0;
    int64_t bp0 = (int64_t)blockIdx.x * kBpPerBlock;
    for (int64_t bp = bp0 + w; bp < bp0 + kBpPerBlock && bp < nbp; bp += 4) {
        uint32_t d[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) {                       // q = 2*e + u
            int64_t t = bp * 8 + (q >> 1) * 4 + h * 2 + (q & 1);
            uint32_t wv = 0;
            if (live) {
                int64_t b0 = t * 4;
                #pragma unroll
                for (int b = 0; b < 4; ++b) if (b0 + b < nbytes) wv |= (uint32_t)src[b0 + b] << (8 * b);
                int64_t rows_left = n - t * 16;
                uint32_t valid = rows_left >= 16 ? 0xFFFFFFFFu : rows_left <= 0 ? 0u : ((1u << (2 * rows_left)) - 1u);
                wv &= valid;
            }
            const uint32_t dd = plink_to_dosage(wv), mm = plink_missing_mask(wv);
            d[q] = dd | mm | (mm << 1);        // missing -> the unused dosage code 3 until k_missing_from_tiles has listed it
            c1 += __popc(dd & 0x55555555u); c2 += __popc(dd & 0xAAAAAAAAu); cm += __popc(mm);
        }
        X[(cg * nbp + bp) * 64 + lane] = make_uint4(d[0], d[1], d[2], d[3]);
    }
    flush_counts(c1, c2, cm, cg, p, cnt);
}

// mu_j = (n1 + 2 n2) / (n - nmiss); sinv_j = 1/sqrt(mu(1-mu/2)) if that sqrt > 0 else 1
__global__ void k_col_stats(const int32_t *__restrict__ cnt, int64_t n, int64_t p,
                            double *__restrict__ mu, double *__restrict__ sinv)
{
    int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= p) return;
    double n1 = cnt[3 * j], n2 = cnt[3 * j + 1], nm = cnt[3 * j + 2];
    double m = (n1 + 2.0 * n2) / ((double)n - nm);
    double s = sqrt(m * (1.0 - m / 2.0));
    mu[j] = m;
    sinv[j] = (s > 0.0) ? 1.0 / s : 1.0;
}

// One wave per column group walks its tiles (one coalesced 1 KB load per 128 rows), appends the rows carrying
// the temporary code 3 to the per-column missing lists in ascending row order and stores the tile back
// with those entries as dosage 0 -- so a .bed is uploaded ONCE (the missing lists used to need a second
// upload of the raw bytes).  Lane 32h+m holds rows 128bp + 64e + 32h + 16u + (0..15) of column m in
// dword 2e+u; the two lanes of a column exchange their per-segment counts to keep the list sorted.
__global__ void __launch_bounds__(64)
k_missing_from_tiles(uint4 *__restrict__ X, int64_t nbp, int64_t p, const int64_t *__restrict__ miss_ptr,
                     int32_t *__restrict__ miss_row)
{
    const int64_t cg = blockIdx.x;
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const int64_t j = cg * 32 + m;
    int64_t base = (j < p) ? miss_ptr[j] : 0;
    const int64_t end = (j < p) ? miss_ptr[j + 1] : 0;
    if (__ballot(end > base) == 0ull) return;                     // no missing entry in this column group
    for (int64_t bp = 0; bp < nbp; ++bp) {
        uint4 v = X[(cg * nbp + bp) * 64 + lane];
        uint32_t d[4] = {v.x, v.y, v.z, v.w}, mk[4];
        int c[2] = {0, 0};
        #pragma unroll
        for (int q = 0; q < 4; ++q) { mk[q] = d[q] & (d[q] >> 1) & 0x55555555u; c[q >> 1] += __popc(mk[q]); }
        const int o0 = __shfl_xor(c[0], 32, 64), o1 = __shfl_xor(c[1], 32, 64);     // the other half's counts
        if (__ballot((c[0] | c[1]) != 0) == 0ull) continue;
        // segment order within a block pair: (e0,h0) (e0,h1) (e1,h0) (e1,h1)
        int64_t w0 = base + (h ? o0 : 0);
        int64_t w1 = base + c[0] + o0 + (h ? o1 : 0);
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            int64_t &w = (q >> 1) ? w1 : w0;
            uint32_t bits = mk[q];
            const int32_t row0 = (int32_t)(bp * 128 + (q >> 1) * 64 + h * 32 + (q & 1) * 16);
            while (bits) {
                int sft = __ffs((int)bits) - 1;
                miss_row[w++] = row0 + (sft >> 1);
                bits &= bits - 1;
            }
            d[q] &= ~(mk[q] | (mk[q] << 1));
        }
        if (c[0] | c[1]) X[(cg * nbp + bp) * 64 + lane] = make_uint4(d[0], d[1], d[2], d[3]);
        base += c[0] + c[1] + o0 + o1;
    }
}

// ---- synthetic SnpArray ------------------------------------------------------
__device__ __forceinline__ void synth_entry(uint32_t key_g, uint32_t key_m, uint32_t thr16,
                                            uint32_t miss_thr, int64_t i, uint32_t &dos, bool &miss)
{
    uint32_t h = lowbias32((uint32_t)i * 0x9E3779B1u ^ key_g);
    dos = ((h & 0xFFFFu) < thr16) + ((h >> 16) < thr16);
    miss = miss_thr != 0 && lowbias32((uint32_t)i * 0x85EBCA77u ^ key_m) < miss_thr;
}

__device__ __forceinline__ void synth_col_keys(uint64_t seed, int64_t j, uint32_t &key_g,
                                               uint32_t &key_m, uint32_t &thr16)
{
    uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    key_g = lowbias32((uint32_t)j ^ lowbias32(s0 ^ 0xA511E9B3u)) ^ lowbias32((uint32_t)(j >> 32) + s1);
    key_m = lowbias32(key_g ^ 0x68E31DA4u);
    uint32_t u = lowbias32(key_g ^ 0xB5297A4Du) >> 8;          // 24 bits
    double maf = 0.5 * ((double)u + 0.5) / 16777216.0;          // U(0, 0.5)
    thr16 = (uint32_t)(maf * 65536.0);
}

__global__ void __launch_bounds__(256)
k_synth(uint4 *__restrict__ X, int64_t nbp, int64_t n, int64_t p, int64_t col0, uint64_t seed,
        uint32_t miss_thr, int32_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    const int64_t cg = blockIdx.y;
    const int64_t j = cg * 32 + m;
    const bool live = j < p;
    uint32_t key_g = 0, key_m = 0, thr16 = 0;
    if (live) synth_col_keys(seed, col0 + j, key_g, key_m, thr16);
    int32_t c1 = 0, c2 = 0, cm = 0;
    int64_t bp0 = (int64_t)blockIdx.x * kBpPerBlock;
    for (int64_t bp = bp0 + w; bp < bp0 + kBpPerBlock && bp < nbp; bp += 4) {
        uint32_t d[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            int64_t t = bp * 8 + (q >> 1) * 4 + h * 2 + (q & 1);
            uint32_t dd = 0;
            if (live) {
                #pragma unroll
                for (int s = 0; s < 16; ++s) {
                    int64_t i = t * 16 + s;
                    uint32_t dos; bool miss;
                    synth_entry(key_g, key_m, thr16, miss_thr, i, dos, miss);
                    if (i < n) {
                        if (miss) cm++;
                        else { dd |= dos << (2 * s); c1 += (dos == 1); c2 += (dos == 2); }
                    }
                }
            }
            d[q] = dd;
        }
        X[(cg * nbp + bp) * 64 + lane] = make_uint4(d[0], d[1], d[2], d[3]);
    }
    flush_counts(c1, c2, cm, cg, p, cnt);
}

__global__ void __launch_bounds__(64)
k_fill_missing_synth(int64_t n, int64_t p, int64_t col0, uint64_t seed, uint32_t miss_thr,
                     const int64_t *__restrict__ miss_ptr, int32_t *__restrict__ miss_row)
{
    int64_t j = blockIdx.x;
    if (j >= p) return;
    uint32_t key_g, key_m, thr16;
    synth_col_keys(seed, col0 + j, key_g, key_m, thr16);
    int64_t base = miss_ptr[j];
    int lane = threadIdx.x;
    for (int64_t i0 = 0; i0 < n; i0 += 64) {
        int64_t i = i0 + lane;
        uint32_t dos; bool miss = false;
        if (i < n) synth_entry(key_g, key_m, thr16, miss_thr, i, dos, miss);
        unsigned long long bal = __ballot(miss);
        if (miss) miss_row[base + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)i;
        base += __popcll(bal);
    }
}

// dense synthetic: approx N(0,1) entries (sum of 4 uniforms, variance-normalised)
__global__ void k_synth_dense(double *__restrict__ D, int64_t total, uint64_t seed)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t s0 = lowbias32((uint32_t)seed ^ 0x3C6EF372u), s1 = lowbias32((uint32_t)(seed >> 32) ^ 0xDAA66D2Bu);
    for (; i < total; i += stride) {
        uint32_t a = lowbias32((uint32_t)i ^ s0) ^ lowbias32((uint32_t)(i >> 32) + s1);
        uint32_t b = lowbias32(a ^ 0x9E3779B9u);
        double u = ((a & 0xFFFF) + (a >> 16) + (b & 0xFFFF) + (b >> 16)) * (1.0 / 65536.0) - 2.0;
        D[i] = u * 1.7320508075688772;   // var of sum of 4 U(0,1) = 1/3
    }
}

// ---- export back to PLINK codes -----------------------------------------------
// out is column-major with `ndw` dwords per column
__global__ void k_export(const uint4 *__restrict__ X, int64_t nbp, int64_t ncg, int64_t p, int64_t ndw,
                         uint32_t *__restrict__ out)
{
    int64_t total = ncg * nbp * 64;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        int lane = (int)(i & 63), m = lane & 31, h = lane >> 5;
        int64_t tile = i >> 6, cg = tile / nbp, bp = tile - cg * nbp, j = cg * 32 + m;
        if (j >= p) continue;
        uint4 v = X[i];
        uint32_t d[4] = {v.x, v.y, v.z, v.w};
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            int64_t t = bp * 8 + (q >> 1) * 4 + h * 2 + (q & 1);
            uint32_t nz = (d[q] | (d[q] >> 1)) & 0x55555555u;   // dosage != 0
            uint32_t two = (d[q] >> 1) & 0x55555555u;           // dosage == 2
            out[j * ndw + t] = (nz << 1) | two;                 // 0->00, 1->10, 2->11
        }
    }
}

__global__ void k_export_missing(uint32_t *__restrict__ out, int64_t ndw, int64_t p,
                                 const int64_t *__restrict__ miss_ptr, const int32_t *__restrict__ miss_row)
{
    int64_t j = blockIdx.x;
    if (j >= p) return;
    for (int64_t t = miss_ptr[j] + threadIdx.x; t < miss_ptr[j + 1]; t += blockDim.x) {
        int32_t i = miss_row[t];
        atomicOr(&out[j * ndw + (i >> 4)], 1u << (2 * (i & 15)));
    }
}

// naive_impute (src/utilities.jl:862-899): the missing entries of a column get the column's most frequent genotype; on a
// tie the reference's if / elseif chain prefers the heterozygote (0x02), then the 0x03 homozygote, then 0x00.
// One wave per column over the exported PLINK codes (missing still 00 here; rows >= n are 00 padding).
__global__ void __launch_bounds__(64)
k_impute_mode(uint32_t *__restrict__ out, int64_t ndw, int64_t p, int64_t n,
              const int64_t *__restrict__ miss_ptr, const int32_t *__restrict__ miss_row)
{
    const int64_t j = blockIdx.x;
    if (j >= p) return;
    const int64_t a = miss_ptr[j], b = miss_ptr[j + 1];
    if (b == a) return;
    long long c1 = 0, c2 = 0;
    for (int64_t t = threadIdx.x; t < ndw; t += 64) {
        const uint32_t d = out[j * ndw + t], hi = (d >> 1) & 0x55555555u, lo = d & 0x55555555u;
        c1 += __popc(hi & ~lo);                          // 10: one copy
        c2 += __popc(hi & lo);                           // 11: two copies
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) { c1 += __shfl_xor(c1, off, 64); c2 += __shfl_xor(c2, off, 64); }
    const long long c0 = n - (b - a) - c1 - c2;
    const long long most = c0 > c1 ? (c0 > c2 ? c0 : c2) : (c1 > c2 ? c1 : c2);
    const uint32_t code = most == c1 ? 2u : most == c2 ? 3u : 0u;
    if (code == 0u) return;
    for (int64_t t = a + threadIdx.x; t < b; t += 64) {
        const int32_t i = miss_row[t];
        atomicOr(&out[j * ndw + (i >> 4)], code << (2 * (i & 15)));
    }
}

static int finish_missing_ptr(mih_mat *h, const std::vector<int32_t> &cnt, std::vector<int64_t> &ptr)
{
    ptr.assign((size_t)h->p + 1, 0);
    for (int64_t j = 0; j < h->p; ++j) ptr[j + 1] = ptr[j] + cnt[3 * j + 2];
    h->total_missing = ptr[h->p];
    MIH_HIP(hipMalloc((void **)&h->miss_ptr, sizeof(int64_t) * (size_t)(h->p + 1)));
    MIH_HIP(hipMemcpy(h->miss_ptr, ptr.data(), sizeof(int64_t) * (size_t)(h->p + 1), hipMemcpyHostToDevice));
    MIH_HIP(hipMalloc((void **)&h->miss_row, sizeof(int32_t) * (size_t)(h->total_missing > 0 ? h->total_missing : 1)));
    return MIH_OK;
}

static int alloc_snp(mih_mat *h)
{
    h->ncg = (h->p + 31) / 32;
    h->nbp = (h->n + 127) / 128;
    h->n_pad = h->nbp * 128;
    size_t bytes = sizeof(uint4) * (size_t)h->ncg * (size_t)h->nbp * 64;
    hipError_t e = hipMalloc((void **)&h->X, bytes);
    if (e != hipSuccess) { set_error("hipMalloc of %zu bytes for the genotype matrix failed: %s", bytes, hipGetErrorString(e)); return MIH_OOM; }
    MIH_HIP(hipMalloc((void **)&h->mu, sizeof(double) * (size_t)h->p));
    MIH_HIP(hipMalloc((void **)&h->sinv, sizeof(double) * (size_t)h->p));
    return MIH_OK;
}

hipStream_t worker_stream(const mih_mat *h, int i)
{
    std::lock_guard<std::mutex> g(h->ws_mu);
    if (i < 0 || i >= 2 * kWorkerStreamsPerLane) return nullptr;
    if ((int)h->worker_streams.size() <= i) h->worker_streams.resize((size_t)i + 1, nullptr);
    if (!h->worker_streams[(size_t)i]) {
        // (round 6, measured and NOT adopted) the greatest stream priority for the per-fit chains of the lock-step lanes: no gain -- a
        // fused pass's workgroups hold every CU until the kernel's last wave of workgroups, whatever the priority of another queue, so
        // the chains run in the windows between two passes either way -- and 0.1 s of 2.4 LOST at configs[3]: with priority the chains
        // of lane A finish before lane B's pass has drained, both lanes queue their passes together, the passes share the CUs and
        // end together, and every round then ends with BOTH lanes' chains in one window (tools/ab_cv_lanes.sh).  Measurement build:
        // MENDELIHT_WORKER_PRIORITY=1 switches it on.
        hipStream_t st = nullptr;
        int least = 0, greatest = 0;
        const bool prio = probe_env("MENDELIHT_WORKER_PRIORITY") != nullptr && atoi(probe_env("MENDELIHT_WORKER_PRIORITY")) != 0;
        if (prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
            if (hipStreamCreateWithPriority(&st, hipStreamDefault, greatest) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
        } else (void)hipGetLastError();
        if (!st && hipStreamCreate(&st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        h->worker_streams[(size_t)i] = st;
    }
    return h->worker_streams[(size_t)i];
}

}  // namespace mih

using namespace mih;

extern "C" {

int mih_device_count(int *count)
{
    if (!count) { set_error("count is NULL"); return MIH_BAD_ARG; }
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return MIH_OK; }
    *count = c;
    return MIH_OK;
}

int mih_last_error(char *buf, size_t len)
{
    if (!buf || len == 0) return MIH_BAD_ARG;
    snprintf(buf, len, "%s", g_err);
    return MIH_OK;
}

int mih_version(int *major, int *minor)
{
    if (major) *major = 0;
    if (minor) *minor = 4;        // round 4: mih_fit_params::cv_threads, mih_cv_allgather, MIH_CNT_INIT_SCORES; round 3: mih_fit_params::xtv_digits, mih_xtv_batched_fmt, mih_profile_* per handle, mih_cv_assignment; mih_set_* gone
    return MIH_OK;
}

int mih_abi_sizes(int64_t *sizes, int32_t n)
{
    const int64_t v[4] = {(int64_t)sizeof(mih_fit_params), (int64_t)sizeof(mih_fit_result), (int64_t)sizeof(mih_mv_result),
                          (int64_t)sizeof(mih_comm)};
    if (!sizes) return MIH_BAD_ARG;
    for (int i = 0; i < n && i < 4; ++i) sizes[i] = v[i];
    return MIH_OK;
}

static int select_device(int device)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c == 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (the MI355X path has no CPU fallback)");
        return MIH_NO_DEVICE;
    }
    if (device < 0 || device >= c) { set_error("device %d out of range (count %d)", device, c); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(device));
    return MIH_OK;
}

// A large matrix reserves the device memory its fits will work in (DevPool, common.h): about what a cross-validation asks for
// -- four fused-pass workspaces and 64 IHTVariable blocks, 15.7 GB beside a 125 GB matrix.  Whatever the driver has to do to
// hand out never-used VRAM (one stall of ~2.9 s was measured) it does here, when the matrix is created, and no fit ever calls
// hipMalloc / hipFree.  The policy is fixed -- a 2-bit matrix of 4 GiB or more gets a reserve when it is created -- and a caller
// changes it with an ARGUMENT, not through the environment: mih_mat_reserve(h, bytes) gives any matrix a reserve
// (bytes = 0: sized by the rule below from its dimensions -- the test suite asks for it on its small matrices so that the pool,
// the arenas and the lock-step hand-over run exactly as beside a 125 GB matrix) or takes it away (bytes < 0).
// MENDELIHT_NO_RESERVE=1 is an A/B switch of the measurement build.
static void reserve_fit_memory(mih_mat *h, bool asked = false, size_t asked_bytes = 0)
{
    static const bool off = probe_env("MENDELIHT_NO_RESERVE") != nullptr;
    const size_t min_bytes = (size_t)(4ull << 30);
    const size_t x_bytes = (size_t)h->ncg * (size_t)h->nbp * 1024;
    if (h->kind != 0 || (!asked && (off || x_bytes < min_bytes))) return;
    size_t big = (size_t)2432 * (size_t)h->p;                    // the row-slice partials of a 19-residual pass (16 slices x 19 x 8 B per column)
    size_t var = 128ull << 20;                                   // one IHTVariable block at n = 500k, p = 1M
    if (x_bytes >= (4ull << 30)) big = std::min<size_t>(std::max<size_t>(big, 256ull << 20), 4ull << 30);
    else {                                                       // a small matrix: the same structure at its own scale
        big = std::max<size_t>(big + (size_t)30 * 8 * (size_t)h->n_pad, 4ull << 20);
        var = std::min<size_t>(var, std::max<size_t>((size_t)8 * (40 * (size_t)h->n_pad + 6 * (size_t)h->p) + (2ull << 20), 4ull << 20));
    }
    const size_t want = asked_bytes ? asked_bytes : 4 * big + 64 * var;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    if (free_b < 4 * want) return;                               // not on a crowded device
    std::shared_ptr<DevPool> pool(new DevPool());
    if (pool->init(want)) { h->pool_owner = pool; h->pool = pool.get(); }
}

int mih_snp_create(const uint8_t *bed_cols, int64_t n, int64_t p, int64_t col_stride_bytes,
                   int center, int scale, int impute, int dtype, int device, mih_mat **out)
{
    if (!bed_cols || !out) { set_error("null argument"); return MIH_BAD_ARG; }
    if (n <= 0 || p <= 0 || col_stride_bytes < (n + 3) / 4) { set_error("bad dimensions n=%lld p=%lld stride=%lld", (long long)n, (long long)p, (long long)col_stride_bytes); return MIH_BAD_DIM; }
    if (n >= (1ll << 31)) { set_error("n must be < 2^31"); return MIH_BAD_DIM; }
    if (dtype != 64 && dtype != 32) { set_error("dtype must be 64 (SnpLinAlg{Float64}) or 32 (SnpLinAlg{Float32})"); return MIH_BAD_ARG; }
    // dtype only records the caller's element type T (src/MendelIHT.jl:39: Float = Union{Float64, Float32}): the 2-bit matrix has no
    // element type on the device and every dot product is exact fixed point recombined in Float64, whatever T is; a binding for
    // T = Float32 converts y, z to Float64 on the way in and the model to Float32 on the way out
    MIH_TRY(select_device(device));
    const bool trace = probe_env("MENDELIHT_INGEST_TRACE") != nullptr;         // measurement build: where the time of a create goes
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = tnow();
    auto lap = [&](const char *what) { if (trace) { const double t = tnow(); fprintf(stderr, "ingest: %-28s %8.2f ms\n", what, t - t_mark); t_mark = t; } };
    mih_mat *h = new mih_mat();
    h->kind = 0; h->device = device; h->n = n; h->p = p;
    h->center = center; h->scale = scale; h->impute = impute;
    int rc = alloc_snp(h);
    if (rc) { mih_mat_destroy(h); return rc; }
    if (hipStreamCreate(&h->stream) != hipSuccess) { mih_mat_destroy(h); return MIH_HIP_ERROR; }
    lap("matrix allocation");

    // Upload pipeline (round 3).  T worker threads, each with its own HIP stream, two pinned staging buffers and two device
    // buffers, pull 16 MB chunks of whole column groups from one queue (small chunks: pinning host memory costs ~0.28 ms per MB, so
    // the 256 MB of staging are 70 ms of every create -- 1 GB of it was 280 ms, more than the transfer of an 8 GB matrix): copy the caller's (pageable, possibly mmapped) columns into
    // the pinned buffer, DMA, transcode -- all on the worker's stream, so the only wait is for the worker's OWN buffer of two
    // chunks ago.  Host copies, PCIe transfers and transcodes of different chunks overlap across the workers; no thread is created
    // per chunk and nothing synchronises the workers with each other.  (Round 2 staged 256 MB chunks through ONE pair of buffers
    // with a fork-join copy per chunk: 34 GB/s against the 56-57 GB/s that tools/ingest_probe.hip measures for every scheme that
    // keeps the link busy -- in-place hipHostRegister in chunks included, so registering the caller's memory buys nothing.)
    auto fail = [&](int code) { mih_mat_destroy(h); return code; };
    const int64_t chunk_bytes = 16ll << 20;
    int64_t cols_per_chunk = chunk_bytes / col_stride_bytes / 32 * 32;   // whole column groups
    if (cols_per_chunk < 32) cols_per_chunk = 32;
    if (cols_per_chunk > round_up(p, 32)) cols_per_chunk = round_up(p, 32);
    const size_t buf_bytes = (size_t)(cols_per_chunk * col_stride_bytes);
    const int64_t nchunks = (p + cols_per_chunk - 1) / cols_per_chunk;
    DevBuf<int32_t> cnt;
    if ((rc = cnt.alloc((size_t)(3 * p)))) return fail(rc);
    if (hipMemsetAsync(cnt.p, 0, sizeof(int32_t) * 3 * (size_t)p, h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    unsigned nth = std::thread::hardware_concurrency();
    nth = nth >= 16 ? 8 : (nth >= 4 ? nth / 2 : 1);
    if (const char *e = probe_env("MENDELIHT_INGEST_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) nth = (unsigned)v; }     // measurement build
    if ((int64_t)nth > nchunks) nth = (unsigned)nchunks;
    //  a chunk is at least one group of 32 columns, so above a 512 KB column stride (n > 2M rows) it outgrows the 16 MB
    // target: n = 40M gives 320 MB chunks, and eight workers with two buffers each would pin 5 GB of host memory (0.28 ms per MB)
    // and take as much VRAM.  The staging of ALL workers stays within a fixed budget: fewer workers for tall matrices
    const size_t staging_budget = 512ull << 20;
    if (buf_bytes * 2 * nth > staging_budget) nth = (unsigned)std::max<size_t>(1, staging_budget / (buf_bytes * 2));
    std::atomic<int64_t> next_chunk{0};
    std::atomic<int> failed{0};
    std::mutex err_mu; std::string err_msg;
    // one pinned block and one device block for all workers (the runtime serialises allocations anyway)
    struct Staging {
        uint8_t *pin = nullptr, *raw = nullptr;
        ~Staging() { if (pin) (void)hipHostFree(pin); if (raw) (void)hipFree(raw); }
    } stg;
    for (;;) {                         // a failed allocation degrades to one worker before it fails the create
        const bool ok = hipHostMalloc((void **)&stg.pin, buf_bytes * 2 * nth, hipHostMallocDefault) == hipSuccess &&
                        hipMalloc((void **)&stg.raw, buf_bytes * 2 * nth) == hipSuccess;
        if (ok) break;
        (void)hipGetLastError();
        if (stg.pin) { (void)hipHostFree(stg.pin); stg.pin = nullptr; }
        if (stg.raw) { (void)hipFree(stg.raw); stg.raw = nullptr; }
        if (nth == 1) { set_error("allocation of the upload staging buffers (2 x %zu bytes pinned + device) failed", buf_bytes); return fail(MIH_OOM); }
        nth = 1;
    }
    lap("staging buffers");
    std::atomic<unsigned> worker_no{0};
    auto worker = [&]() {
        const unsigned me = worker_no.fetch_add(1);
        struct Res {
            uint8_t *pin[2] = {nullptr, nullptr}, *raw[2] = {nullptr, nullptr}; hipStream_t st = nullptr; hipEvent_t done[2] = {nullptr, nullptr};
            ~Res() {
                if (st) (void)hipStreamSynchronize(st);
                for (int i = 0; i < 2; ++i) if (done[i]) (void)hipEventDestroy(done[i]);
                if (st) (void)hipStreamDestroy(st);
            }
        } r;
        auto bad = [&](const char *what) { std::lock_guard<std::mutex> g(err_mu); if (err_msg.empty()) err_msg = what; failed.store(1); };
        if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&r.st) != hipSuccess) return bad("ingest worker: stream");
        for (int i = 0; i < 2; ++i) {
            r.pin[i] = stg.pin + buf_bytes * (2 * me + i);
            r.raw[i] = stg.raw + buf_bytes * (2 * me + i);
            if (hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming) != hipSuccess) return bad("ingest worker: event");
        }
        for (int64_t it = 0; !failed.load(); ++it) {
            const int64_t c = next_chunk.fetch_add(1);
            if (c >= nchunks) break;
            const int b = (int)(it & 1);
            if (it >= 2 && hipEventSynchronize(r.done[b]) != hipSuccess) return bad("ingest worker: wait");      // buffers b are free again
            const int64_t c0 = c * cols_per_chunk, nc = std::min<int64_t>(cols_per_chunk, p - c0);
            const size_t bytes = (size_t)(nc * col_stride_bytes);
            std::memcpy(r.pin[b], bed_cols + c0 * col_stride_bytes, bytes);
            if (hipMemcpyAsync(r.raw[b], r.pin[b], bytes, hipMemcpyHostToDevice, r.st) != hipSuccess) return bad("ingest worker: H2D copy");
            dim3 grid((unsigned)((h->nbp + kBpPerBlock - 1) / kBpPerBlock), (unsigned)((nc + 31) / 32));
            hipLaunchKernelGGL(k_transcode, grid, dim3(256), 0, r.st, r.raw[b], col_stride_bytes, n, p, c0, nc,
                               reinterpret_cast<uint4 *>(h->X), h->nbp, cnt.p);
            if (hipEventRecord(r.done[b], r.st) != hipSuccess) return bad("ingest worker: event record");
        }
        if (hipStreamSynchronize(r.st) != hipSuccess) return bad("transcode kernel failed");
    };
    if (nth <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nth; ++t) th.emplace_back(worker);
        for (auto &t : th) t.join();
    }
    if (failed.load()) { set_error("%s", err_msg.c_str()); (void)hipGetLastError(); return fail(MIH_HIP_ERROR); }
    lap("copy + DMA + transcode");
    if (hipStreamSynchronize(h->stream) != hipSuccess) { set_error("transcode kernel failed"); return fail(MIH_HIP_ERROR); }
    hipLaunchKernelGGL(k_col_stats, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, h->stream, cnt.p, n, p, h->mu, h->sinv);
    std::vector<int32_t> hcnt((size_t)(3 * p));
    if (hipMemcpy(hcnt.data(), cnt.p, sizeof(int32_t) * hcnt.size(), hipMemcpyDeviceToHost) != hipSuccess) return fail(MIH_HIP_ERROR);
    std::vector<int64_t> ptr;
    if ((rc = finish_missing_ptr(h, hcnt, ptr))) return fail(rc);
    if (h->total_missing > 0)
        hipLaunchKernelGGL(k_missing_from_tiles, dim3((unsigned)h->ncg), dim3(64), 0, h->stream, reinterpret_cast<uint4 *>(h->X), h->nbp, p,
                           h->miss_ptr, h->miss_row);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    lap("statistics + missing lists");
    reserve_fit_memory(h);
    lap("reserve for the fits");
    *out = h;
    return MIH_OK;
}

int mih_snp_create_synthetic(int64_t n, int64_t p, uint64_t seed, double missing_rate,
                             int center, int scale, int impute, int device, mih_mat **out)
{
    return mih_snp_create_synthetic_shard(n, p, 0, seed, missing_rate, center, scale, impute, device, out);
}

int mih_snp_create_synthetic_shard(int64_t n, int64_t p, int64_t col_offset, uint64_t seed, double missing_rate,
                                   int center, int scale, int impute, int device, mih_mat **out)
{
    if (!out) return MIH_BAD_ARG;
    if (n <= 0 || p <= 0 || n >= (1ll << 31)) { set_error("bad dimensions"); return MIH_BAD_DIM; }
    if (!(missing_rate >= 0.0 && missing_rate < 1.0)) { set_error("missing_rate must be in [0,1)"); return MIH_BAD_ARG; }
    MIH_TRY(select_device(device));
    mih_mat *h = new mih_mat();
    h->kind = 0; h->device = device; h->n = n; h->p = p;
    h->center = center; h->scale = scale; h->impute = impute;
    int rc = alloc_snp(h);
    if (rc) { mih_mat_destroy(h); return rc; }
    auto fail = [&](int code) { mih_mat_destroy(h); return code; };
    if (hipStreamCreate(&h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    DevBuf<int32_t> cnt;
    if ((rc = cnt.alloc((size_t)(3 * p)))) return fail(rc);
    uint32_t miss_thr = (uint32_t)(missing_rate * 4294967296.0);
    if (hipMemsetAsync(cnt.p, 0, sizeof(int32_t) * 3 * (size_t)p, h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    dim3 grid((unsigned)((h->nbp + kBpPerBlock - 1) / kBpPerBlock), (unsigned)h->ncg);
    hipLaunchKernelGGL(k_synth, grid, dim3(256), 0, h->stream, reinterpret_cast<uint4 *>(h->X), h->nbp, n, p, col_offset, seed, miss_thr, cnt.p);
    hipLaunchKernelGGL(k_col_stats, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, h->stream, cnt.p, n, p, h->mu, h->sinv);
    if (hipStreamSynchronize(h->stream) != hipSuccess) { set_error("synthetic generator failed: %s", hipGetErrorString(hipGetLastError())); return fail(MIH_HIP_ERROR); }
    std::vector<int32_t> hcnt((size_t)(3 * p));
    if (hipMemcpy(hcnt.data(), cnt.p, sizeof(int32_t) * hcnt.size(), hipMemcpyDeviceToHost) != hipSuccess) return fail(MIH_HIP_ERROR);
    std::vector<int64_t> ptr;
    if ((rc = finish_missing_ptr(h, hcnt, ptr))) return fail(rc);
    if (h->total_missing > 0) {
        hipLaunchKernelGGL(k_fill_missing_synth, dim3((unsigned)p), dim3(64), 0, h->stream, n, p, col_offset, seed, miss_thr, h->miss_ptr, h->miss_row);
        if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    }
    reserve_fit_memory(h);
    *out = h;
    return MIH_OK;
}

int mih_dense_create(const double *x, int64_t n, int64_t p, int device, mih_mat **out)
{
    if (!x || !out) return MIH_BAD_ARG;
    if (n <= 0 || p <= 0) return MIH_BAD_DIM;
    MIH_TRY(select_device(device));
    mih_mat *h = new mih_mat();
    h->kind = 1; h->device = device; h->n = n; h->p = p; h->center = h->scale = h->impute = 0;
    auto fail = [&](int code) { mih_mat_destroy(h); return code; };
    if (hipMalloc((void **)&h->D, sizeof(double) * (size_t)n * (size_t)p) != hipSuccess) { set_error("hipMalloc for dense matrix failed"); (void)hipGetLastError(); return fail(MIH_OOM); }
    if (hipMemcpy(h->D, x, sizeof(double) * (size_t)n * (size_t)p, hipMemcpyHostToDevice) != hipSuccess) return fail(MIH_HIP_ERROR);
    if (hipStreamCreate(&h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    *out = h;
    return MIH_OK;
}

int mih_dense_create_f32(const float *x, int64_t n, int64_t p, int device, mih_mat **out)
{
    if (!x || !out) return MIH_BAD_ARG;
    if (n <= 0 || p <= 0) return MIH_BAD_DIM;
    MIH_TRY(select_device(device));
    mih_mat *h = new mih_mat();
    h->kind = 1; h->device = device; h->n = n; h->p = p; h->center = h->scale = h->impute = 0;
    auto fail = [&](int code) { mih_mat_destroy(h); return code; };
    if (hipMalloc((void **)&h->Df, sizeof(float) * (size_t)n * (size_t)p) != hipSuccess) { set_error("hipMalloc for dense matrix failed"); (void)hipGetLastError(); return fail(MIH_OOM); }
    if (hipMemcpy(h->Df, x, sizeof(float) * (size_t)n * (size_t)p, hipMemcpyHostToDevice) != hipSuccess) return fail(MIH_HIP_ERROR);
    if (hipStreamCreate(&h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    *out = h;
    return MIH_OK;
}

int mih_dense_create_synthetic(int64_t n, int64_t p, uint64_t seed, int device, mih_mat **out)
{
    if (!out) return MIH_BAD_ARG;
    if (n <= 0 || p <= 0) return MIH_BAD_DIM;
    MIH_TRY(select_device(device));
    mih_mat *h = new mih_mat();
    h->kind = 1; h->device = device; h->n = n; h->p = p; h->center = h->scale = h->impute = 0;
    auto fail = [&](int code) { mih_mat_destroy(h); return code; };
    if (hipMalloc((void **)&h->D, sizeof(double) * (size_t)n * (size_t)p) != hipSuccess) { set_error("hipMalloc for dense matrix failed"); (void)hipGetLastError(); return fail(MIH_OOM); }
    if (hipStreamCreate(&h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    hipLaunchKernelGGL(k_synth_dense, dim3(8192), dim3(256), 0, h->stream, h->D, n * p, seed);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(MIH_HIP_ERROR);
    *out = h;
    return MIH_OK;
}

int mih_mat_destroy(mih_mat *h)
{
    if (!h) return MIH_OK;
    (void)hipSetDevice(h->device);
    for (hipStream_t ws : h->worker_streams) if (ws) { (void)hipStreamSynchronize(ws); (void)hipStreamDestroy(ws); }
    h->worker_streams.clear();
    if (h->X) (void)hipFree(h->X);
    if (h->mu) (void)hipFree(h->mu);
    if (h->sinv) (void)hipFree(h->sinv);
    if (h->miss_ptr) (void)hipFree(h->miss_ptr);
    if (h->miss_row) (void)hipFree(h->miss_row);
    if (h->D) (void)hipFree(h->D);
    if (h->Df) (void)hipFree(h->Df);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;                                  // the reserve goes with its last owner (pool_owner)
    return MIH_OK;
}

int mih_mat_reserve(mih_mat *h, int64_t bytes)
{
    if (!h) { set_error("null handle"); return MIH_BAD_ARG; }
    if (h->kind != 0) return MIH_OK;                       // dense matrices keep no reserve
    MIH_HIP(hipSetDevice(h->device));
    if (bytes < 0) { h->pool = nullptr; h->pool_owner.reset(); return MIH_OK; }      // (a running session keeps its own reference)
    if (h->pool) return MIH_OK;                            // it has one already
    reserve_fit_memory(h, true, (size_t)bytes);
    if (!h->pool) { set_error("the device has no room for a reserve of the asked size beside this matrix"); return MIH_OOM; }
    return MIH_OK;
}

int mih_mat_dims(const mih_mat *h, int64_t *n, int64_t *p)
{
    if (!h) return MIH_BAD_ARG;
    if (n) *n = h->n;
    if (p) *p = h->p;
    return MIH_OK;
}

int mih_snp_mu_sigma(const mih_mat *h, double *mu, double *sinv)
{
    if (!h || h->kind != 0) { set_error("not a SnpLinAlg handle"); return MIH_BAD_ARG; }
    MIH_HIP(hipSetDevice(h->device));
    if (mu) MIH_HIP(hipMemcpy(mu, h->mu, sizeof(double) * (size_t)h->p, hipMemcpyDeviceToHost));
    if (sinv) MIH_HIP(hipMemcpy(sinv, h->sinv, sizeof(double) * (size_t)h->p, hipMemcpyDeviceToHost));
    return MIH_OK;
}

static int export_codes(const mih_mat *h, uint8_t *bed_cols_out, bool impute_mode);
int mih_snp_export_bed(const mih_mat *h, uint8_t *bed_cols_out) { return export_codes(h, bed_cols_out, false); }
int mih_snp_naive_impute(const mih_mat *h, uint8_t *bed_cols_out) { return export_codes(h, bed_cols_out, true); }
static int export_codes(const mih_mat *h, uint8_t *bed_cols_out, bool impute_mode)
{
    if (!h || h->kind != 0 || !bed_cols_out) return MIH_BAD_ARG;
    MIH_HIP(hipSetDevice(h->device));
    DevBuf<uint32_t> tmp;
    int64_t ndw = h->n_pad / 16;
    MIH_TRY(tmp.alloc((size_t)h->p * (size_t)ndw));
    hipLaunchKernelGGL(k_export, dim3(4096), dim3(256), 0, h->stream, reinterpret_cast<const uint4 *>(h->X), h->nbp, h->ncg, h->p, ndw, tmp.p);
    if (h->total_missing > 0 && impute_mode)
        hipLaunchKernelGGL(k_impute_mode, dim3((unsigned)h->p), dim3(64), 0, h->stream, tmp.p, ndw, h->p, h->n, h->miss_ptr, h->miss_row);
    else if (h->total_missing > 0)
        hipLaunchKernelGGL(k_export_missing, dim3((unsigned)h->p), dim3(64), 0, h->stream, tmp.p, ndw, h->p, h->miss_ptr, h->miss_row);
    MIH_HIP(hipStreamSynchronize(h->stream));
    size_t width = (size_t)((h->n + 3) / 4);
    MIH_HIP(hipMemcpy2D(bed_cols_out, width, tmp.p, (size_t)ndw * 4, width, (size_t)h->p, hipMemcpyDeviceToHost));
    return MIH_OK;
}

}  // extern "C"
