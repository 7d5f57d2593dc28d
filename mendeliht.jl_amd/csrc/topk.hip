// topk.hip -- project_k!(x, k) on the device (src/utilities.jl:553-559): a = |k-th largest by magnitude|, then every |x_i| < a is
// zeroed; entries tied with a are KEPT.  Keys are the IEEE-754 bit patterns of |x| (monotone for non-negative doubles, +Inf sorts
// last so the `zkeep` slots of vectorize! (utilities.jl:313-314) always survive); all counting is in integers, so the selected
// support is exact and reproducible.
//   * the stand-alone / host-driven select (topk_project_device): TWO histogram sweeps of 11 bits each (exponent, then the top
//     mantissa bits: k_hist11 / k_pick11) pin the threshold down to a 22-bit prefix, k_collect gathers everything at or above
//     that prefix, and the host finishes among the few entries that share it (std::nth_element on the keys, ties kept).  Eight
//     sweeps of 8 bits (k_hist / k_pick / k_threshold) remain as the fallback for massive ties (more prefix-sharers than the
//     gather buffer holds).
//   * a device-resident fit (fit_state.h, resident.inc) does NOT come here: its select -- the same two sweeps or a verified direct
//     gather, and the exact finish by rank counting in one workgroup -- never leaves the device (k_res_grad .. k_res_select).
// Also here: the polled readback (SpinFlag / k_publish) and the pinned upload ring (HostStage / k_stage) of the host-driven step.
#include "common.h"
#include <chrono>
#include <algorithm>
#include <cstring>
#include <functional>

namespace mih {

__device__ __forceinline__ uint64_t abs_key(double v)
{
    return (uint64_t)__double_as_longlong(v) & 0x7FFFFFFFFFFFFFFFull;
}

// pick the bin holding the kth largest: 256 threads, suffix sums in LDS.  (Folding this into the last
// workgroup of k_hist with a ticket counter was measured SLOWER: the agent-scope fence it needs writes
// back the XCD's L2 and cost ~30 us per pass, against ~5 us for this separate launch.)
__global__ void __launch_bounds__(256)
k_pick(uint32_t *__restrict__ hist, uint64_t *__restrict__ state, int shift, uint64_t *__restrict__ sel)
{
    __shared__ uint64_t suf[257];      // suf[b] = sum of hist[b..255]
    __shared__ int chosen;
    const int b = threadIdx.x;
    suf[b] = hist[b];
    if (b == 0) { suf[256] = 0; chosen = 0; }
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint64_t add = (b + off < 256) ? suf[b + off] : 0;
        __syncthreads();
        suf[b] += add;
        __syncthreads();
    }
    const uint64_t kth = state[1];
    // the bin is the largest b with suf[b] >= kth (bin 0 if none)
    if (b > 0 && suf[b] >= kth && suf[b + 1] < kth) chosen = b;
    __syncthreads();
    hist[b] = 0;
    if (b == 0) {
        const int bin = chosen;
        state[0] = (shift == 56 ? 0ull : (state[0] << 8)) | (uint64_t)bin;
        state[1] = kth - suf[bin + 1];
        if (shift == 0) { state[2] = state[0]; sel[0] = 0; sel[1] = 0; }   // full 64-bit threshold key; survivor counter
    }
}

// state[0] = prefix (bits above `shift+8` already fixed), state[1] = remaining rank.
// |x| values of a gradient cluster in a few exponent bins, so plain LDS atomics serialise: every wave
// has a private histogram and adds one count per distinct bin among its 64 lanes (ballot aggregation).
__global__ void __launch_bounds__(256)
k_hist(const double *__restrict__ x, int64_t len, int shift, const uint64_t *__restrict__ state,
       uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[4][256];
    for (int e = threadIdx.x; e < 4 * 256; e += 256) (&h[0][0])[e] = 0;
    __syncthreads();
    const uint64_t prefix = state[0];
    const bool first = (shift == 56);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (len + stride - 1) / stride;             // every lane runs every round (ballots)
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t r = 0; r < rounds; ++r, i += stride) {
        bool live = false; uint32_t bin = 0;
        if (i < len) {
            uint64_t key = abs_key(x[i]);
            live = first || (key >> (shift + 8)) == prefix;
            bin = (uint32_t)(key >> shift) & 255u;
        }
        if (!first) {                       // lower bytes are spread over many bins: plain LDS atomics are cheapest
            if (live) atomicAdd(&h[wave][bin], 1u);
            continue;
        }
        uint64_t todo = __ballot(live);     // exponent byte: a handful of bins per wave -> one add per distinct bin
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lb = __shfl(bin, leader, 64);
            const uint64_t same = __ballot(live && bin == lb) & todo;
            if (lane == leader) h[wave][lb] += (uint32_t)__popcll(same);   // one lane per wave-private bin: no atomic needed
            todo &= ~same;
        }
    }
    __syncthreads();
    const uint32_t tot = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
    if (tot) atomicAdd(&hist[threadIdx.x], tot);
}

// sel[0] = count; survivor t is the pair sel[2 + 2t] = index, sel[3 + 2t] = value bits
__global__ void __launch_bounds__(256)
k_threshold(double *__restrict__ x, int64_t len, const uint64_t *__restrict__ state, uint64_t *__restrict__ sel, uint32_t cap)
{
    const uint64_t thr = state[2];
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < len; i += stride) {
        double v = x[i];
        if (abs_key(v) < thr) { if (v != 0.0) x[i] = 0.0; }
        else if (v != 0.0) {
            unsigned long long pos = atomicAdd((unsigned long long *)sel, 1ull);
            if (pos < cap) { sel[2 + 2 * pos] = (uint64_t)i; sel[3 + 2 * pos] = (uint64_t)__double_as_longlong(v); }
        }
    }
}

// ---- two passes of 11 bits, then the host finishes --------------------------------------------------------------------
// The k-th largest magnitude is pinned down to its exponent and its top 11 mantissa bits by two histogram passes (2048
// bins each); everything at or above that 22-bit prefix -- the entries certainly kept plus the handful sharing the prefix
// with the threshold -- is gathered in one more sweep, and the host picks the exact threshold among the prefix-sharers
// (ties kept, utilities.jl:553-559).  5 launches instead of 17; the 8 x 8-bit select above remains the fallback when the
// prefix is shared by more entries than the gather buffer holds (massive ties).
constexpr int kBins11 = 2048;
__global__ void __launch_bounds__(256)
k_hist11(const double *__restrict__ x, int64_t len, int second, const uint64_t *__restrict__ state, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[4][kBins11];
    for (int e = threadIdx.x; e < 4 * kBins11; e += 256) (&h[0][0])[e] = 0;
    __syncthreads();
    const uint64_t prefix = state[0];
    const int wave = threadIdx.x >> 6;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    // four independent loads in flight per thread; plain LDS adds into the wave's own copy (a wave's 64 keys fall into a
    // handful of exponent bins: the LDS serialises those adds faster than a ballot loop over the distinct bins does)
    for (; i + 3 * stride < len; i += 4 * stride) {
        uint64_t key[4];
        #pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = abs_key(x[i + u * stride]);
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!second) atomicAdd(&h[wave][(uint32_t)(key[u] >> 52)], 1u);
            else if ((key[u] >> 52) == prefix) atomicAdd(&h[wave][(uint32_t)(key[u] >> 41) & 2047u], 1u);
        }
    }
    for (; i < len; i += stride) {
        const uint64_t key = abs_key(x[i]);
        if (!second) atomicAdd(&h[wave][(uint32_t)(key >> 52)], 1u);
        else if ((key >> 52) == prefix) atomicAdd(&h[wave][(uint32_t)(key >> 41) & 2047u], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kBins11; b += 256) {
        const uint32_t tot = h[0][b] + h[1][b] + h[2][b] + h[3][b];
        if (tot) atomicAdd(&hist[b], tot);
    }
}
// pick the bin holding the kth largest among 2048: thread t owns bins 8t .. 8t+7.  second = 0: kth comes as an argument
// (no host-to-device copy of the state); second = 1: also arms the gather (lower-bound key, remaining rank, counter).
__global__ void __launch_bounds__(256)
k_pick11(uint32_t *__restrict__ hist, uint64_t *__restrict__ state, int second, uint64_t k_arg, uint64_t *__restrict__ sel)
{
    __shared__ uint64_t suf[257];      // suf[t] = sum of the bins of threads t..255
    const int t = threadIdx.x;
    uint32_t mine[8]; uint64_t own = 0;
    #pragma unroll
    for (int j = 0; j < 8; ++j) { mine[j] = hist[8 * t + j]; own += mine[j]; hist[8 * t + j] = 0; }
    suf[t] = own;
    if (t == 0) suf[256] = 0;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint64_t add = (t + off < 256) ? suf[t + off] : 0;
        __syncthreads();
        suf[t] += add;
        __syncthreads();
    }
    const uint64_t kth = second ? state[1] : k_arg;
    // the owning thread: entries above my bins < kth <= entries at or above my bins (thread 0 takes it if none does)
    const bool owner = (suf[t + 1] < kth && suf[t] >= kth) || (t == 0 && suf[0] < kth);
    if (owner) {
        uint64_t above = suf[t + 1];
        int bin = 8 * t;
        for (int j = 7; j >= 0; --j) { if (above + mine[j] >= kth) { bin = 8 * t + j; break; } above += mine[j]; }
        if (suf[0] < kth) { bin = 0; above = suf[0] - mine[0]; }      // fewer than kth entries in all: the threshold is the lowest bin
        const uint64_t pre = second ? ((state[0] << 11) | (uint64_t)bin) : (uint64_t)bin;
        state[0] = pre;
        state[1] = kth > above ? kth - above : 1;
        if (second) { state[2] = pre << 41; sel[0] = 0; sel[1] = 0; }
    }
}
// everything at or above the 22-bit prefix: sel[0] = count, pairs (index, value bits) behind it
__global__ void __launch_bounds__(256)
k_collect(const double *__restrict__ x, int64_t len, const uint64_t *__restrict__ state, uint64_t *__restrict__ sel, uint32_t cap)
{
    const uint64_t lo = state[2];
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < len; i += stride) {
        const double v = x[i];
        if (abs_key(v) >= lo && v != 0.0) {
            unsigned long long pos = atomicAdd((unsigned long long *)sel, 1ull);
            if (pos < cap) { sel[2 + 2 * pos] = (uint64_t)i; sel[3 + 2 * pos] = (uint64_t)__double_as_longlong(v); }
        }
    }
}
__global__ void __launch_bounds__(256)
k_zero_below(double *__restrict__ x, int64_t len, uint64_t thr)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < len; i += stride) { const double v = x[i]; if (abs_key(v) < thr && v != 0.0) x[i] = 0.0; }
}

// (round 6) project_k!'s exact finish ON THE DEVICE (utilities.jl:553-559: |x_i| >= the K-th largest magnitude survives, ties kept):
// one workgroup ranks the gathered candidates -- every entry at or above the threshold's 22-bit prefix, so the K largest of the
// vector are among them -- by counting (an entry survives iff fewer than K candidates are strictly larger: any order of counting
// gives the same set), orders the survivors by index and leaves `count, threshold key, (index, value bits) ...` for the way home.
// Rounds 1-5 brought the candidates home and ran std::nth_element + std::sort on the host.  out[0] = ~0: more candidates than this
// kernel ranks (massive ties): the host finishes as before.
constexpr int kFinishCap = 4096;
constexpr int kFinishBin = 1024;            // candidates sharing the threshold's 22-bit prefix that it ranks among themselves
__global__ void __launch_bounds__(1024)
k_topk_finish(const uint64_t *__restrict__ sel, uint32_t cap, uint64_t k, const uint64_t *__restrict__ state, uint64_t *__restrict__ out)
{
    __shared__ uint64_t bits[kFinishCap]; __shared__ int64_t idx[kFinishCap]; __shared__ int32_t keep[kFinishCap];
    __shared__ uint64_t bkey[kFinishBin]; __shared__ int32_t bpos[kFinishBin];
    __shared__ int ns, nbin; __shared__ unsigned long long kmin;
    const int tid = threadIdx.x;
    const uint64_t cnt64 = sel[0];
    if (cnt64 > (uint64_t)kFinishCap || cnt64 > (uint64_t)cap) { if (tid == 0) { out[0] = ~0ull; out[1] = 0; } return; }
    const int cnt = (int)cnt64;
    // what the two histogram passes left (k_pick11): the threshold's 22-bit prefix and its rank among the entries that share it
    const uint64_t pre = state[0], rem = state[1], M = 0x7FFFFFFFFFFFFFFFull;
    if (tid == 0) { ns = 0; nbin = 0; kmin = ~0ull; }
    __syncthreads();
    for (int e = tid; e < cnt; e += 1024) {
        const int64_t je = (int64_t)sel[2 + 2 * e]; const uint64_t be = sel[3 + 2 * e];
        idx[e] = je; bits[e] = be;
        const uint64_t top = (be & M) >> 41;
        keep[e] = top > pre ? 1 : 0;                         // above the prefix: among the K largest for certain
        if (top == pre) { const int q = atomicAdd(&nbin, 1); if (q < kFinishBin) { bkey[q] = be & M; bpos[q] = e; } }
    }
    __syncthreads();
    const int nb = nbin;
    if (nb > kFinishBin) { if (tid == 0) { out[0] = ~0ull; out[1] = 0; } return; }      // (uniform: massive ties, the host finishes)
    // the sharers among themselves: an entry survives iff fewer than `rem` of them are strictly larger (ties at the threshold are
    // kept); fewer than K non-zeros in all: everything gathered survives
    for (int q = tid; q < nb; q += 1024) {
        const uint64_t kq = bkey[q];
        uint64_t larger = 0;
        for (int f = 0; f < nb; ++f) larger += bkey[f] > kq;
        if ((uint64_t)cnt < k || larger < rem) keep[bpos[q]] = 1;
    }
    __syncthreads();
    // order by index: T threads per candidate, each counting over its share of the list (a count: any order)
    int T = 1;
    while (T < 64 && 2 * T * cnt <= 1024) T *= 2;
    const int share = (cnt + T - 1) / T;
    for (int base = 0; base < cnt; base += 1024 / T) {
        const int e = base + tid / T, part = tid % T;
        const bool on = e < cnt && keep[e];
        const int64_t je = on ? idx[e] : 0;
        int pos = 0;
        if (on) { const int f1 = (part + 1) * share < cnt ? (part + 1) * share : cnt; for (int f = part * share; f < f1; ++f) pos += (keep[f] && idx[f] < je) ? 1 : 0; }
        for (int off = 1; off < T; off <<= 1) pos += __shfl_xor(pos, off, 64);
        if (on && part == 0) {
            out[2 + 2 * pos] = (uint64_t)je; out[3 + 2 * pos] = bits[e];
            atomicAdd(&ns, 1);
            if ((uint64_t)cnt >= k) atomicMin(&kmin, (unsigned long long)(bits[e] & M));
        }
    }
    __syncthreads();
    if (tid == 0) { out[0] = (uint64_t)ns; out[1] = (uint64_t)cnt >= k ? (uint64_t)kmin : 0ull; }
}
// x_i with |x_i| below the threshold k_topk_finish left -> 0
__global__ void __launch_bounds__(256)
k_zero_below_dev(double *__restrict__ x, int64_t len, const uint64_t *__restrict__ fin)
{
    if (fin[0] == ~0ull) return;
    const uint64_t thr = fin[1];
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < len; i += stride) { const double v = x[i]; if (abs_key(v) < thr && v != 0.0) x[i] = 0.0; }
}

__global__ void __launch_bounds__(256)
k_publish(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst_host, uint64_t words, long long pairs_first,
          uint64_t *__restrict__ flag_host, uint64_t seq)
{
    if (pairs_first >= 0) {
        const uint64_t cnt = src[0];
        words = 2 + 2 * (cnt < (uint64_t)pairs_first ? cnt : (uint64_t)pairs_first);
    }
    publish_block(src, dst_host, words, flag_host, seq);
}

// how long a lane's fit polls (switching to the lane's other fits in between) before the thread goes to sleep in hipStreamSynchronize
static long coop_spin_us()
{
    static const long us = [] { const char *e = probe_env("MENDELIHT_COOP_SPIN_US"); return e ? atol(e) : 300l; }();
    return us;
}
static bool spin_on()
{
    static const bool on = [] { const char *e = probe_env("MENDELIHT_NO_SPIN"); return !(e && atoi(e) != 0); }();
    return on;
}
uint64_t spin_begin(SpinFlag &f)
{
    if (!spin_on()) return 0;
    if (!f.word.p) { if (f.word.alloc(8, true)) return 0; f.word.p[0] = 0; f.seq = 0; }
    return ++f.seq;
}
int stream_sync_coop(hipStream_t s)
{
    if (coop_can_yield()) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 0;; ++it) {
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) return MIH_OK;
            if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery", __FILE__, __LINE__);
            current_coop()->yield();
            if ((it & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(coop_spin_us())) break;      // see spin_wait
        }
    }
    MIH_HIP(hipStreamSynchronize(s));
    return MIH_OK;
}
int spin_wait(hipStream_t s, SpinFlag &f, uint64_t seq)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (coop_can_yield()) {
        // a lane's fit: let the thread queue the other fits' chains while this readback is on its way; when every fit of the lane
        // waits, the scheduler's round-robin is the spin.  A readback that is not there after 300 us sits behind a fused pass of
        // tens of ms (every fit of the lane then waits for the same pass): stop switching contexts -- every switch is two
        // sigprocmask system calls -- and let the thread sleep in the synchronise below.
        for (unsigned it = 0;; ++it) {
            if (__atomic_load_n(f.word.p, __ATOMIC_ACQUIRE) == seq) return MIH_OK;
            current_coop()->yield();
            if ((it & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(coop_spin_us())) break;
        }
    }
    else for (unsigned it = 0;; ++it) {
        if (__atomic_load_n(f.word.p, __ATOMIC_ACQUIRE) == seq) return MIH_OK;
        __builtin_ia32_pause();
        if ((it & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
    }
    MIH_HIP(hipStreamSynchronize(s));          // a long kernel sits in front of the chain (or a launch failed: reported here)
    if (__atomic_load_n(f.word.p, __ATOMIC_ACQUIRE) != seq) { set_error("readback kernel did not complete"); return MIH_HIP_ERROR; }
    return MIH_OK;
}
int readback_words(hipStream_t s, SpinFlag &f, const uint64_t *src_dev, uint64_t *dst_host, size_t words, int64_t pairs_first)
{
    const uint64_t seq = words ? spin_begin(f) : 0;
    if (!seq) {
        if (words) MIH_HIP(hipMemcpyAsync(dst_host, src_dev, sizeof(uint64_t) * words, hipMemcpyDeviceToHost, s));
        MIH_HIP(hipStreamSynchronize(s));
        return MIH_OK;
    }
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, s, src_dev, dst_host, (uint64_t)words, (long long)pairs_first, f.word.p, seq);
    return spin_wait(s, f, seq);
}

__global__ void __launch_bounds__(256)
k_stage(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst_a, uint64_t words_a, uint64_t *__restrict__ dst_b, uint64_t words_b)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i < words_a) dst_a[i] = src[i];
    else if (i < words_a + words_b) dst_b[i - words_a] = src[i];
}
void stage_to_device(hipStream_t s, const uint64_t *src_pinned, uint64_t *dst_a, size_t words_a, uint64_t *dst_b, size_t words_b)
{
    const size_t tot = words_a + words_b;
    if (tot) hipLaunchKernelGGL(k_stage, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src_pinned, dst_a, (uint64_t)words_a, dst_b, (uint64_t)words_b);
}
int HostStage::put(hipStream_t s, const void *a, size_t bytes_a, const void *b, size_t bytes_b, const uint64_t **out)
{
    *out = nullptr;
    const size_t wa = (bytes_a + 7) / 8, wb = (bytes_b + 7) / 8;
    if (!ring.p || wa + wb > slot_words) return MIH_OK;
    if (since_sync >= kSlots - 1) { MIH_HIP(hipStreamSynchronize(s)); since_sync = 0; }     // the slot may still be unread
    uint64_t *dst = ring.p + (size_t)next * slot_words;
    if (bytes_a) std::memcpy(dst, a, bytes_a);
    if (bytes_b) std::memcpy(dst + wa, b, bytes_b);
    next = (next + 1) % kSlots; ++since_sync;
    *out = dst;
    return MIH_OK;
}

int topk_work_init(TopkWork &w, int64_t max_keep)
{
    MIH_TRY(w.hist.alloc(kBins11));
    MIH_HIP(hipMemset(w.hist.p, 0, kBins11 * sizeof(uint32_t)));  // the pick kernels leave the histogram zeroed after every pass
    if (const char *e = probe_env("MENDELIHT_TOPK_RADIX8")) w.radix8 = atoi(e) != 0;
    MIH_TRY(w.state.alloc(4));
    w.expect = max_keep + 64;
    w.cap = max_keep + 1024;
    MIH_TRY(w.sel.alloc(2 + 2 * (size_t)w.cap));
    MIH_TRY(w.fin.alloc(2 + 2 * (size_t)kFinishCap));
    MIH_TRY(w.hsel.alloc(2 + 2 * (size_t)w.expect, true));
    MIH_TRY(w.flag.word.alloc(8, true)); w.flag.word.p[0] = 0; w.flag.seq = 0;
    return MIH_OK;
}

// threshold pass with the key already in state[2]: zero what is below, gather what survives.  The count
// and the first `expect` pairs come back in ONE device-to-host copy (one host synchronisation).
static int compact_device(double *x_dev, int64_t len, TopkWork &w, hipStream_t s,
                          std::vector<int64_t> &idx_out, std::vector<double> &val_out, bool counter_cleared = false)
{
    int grid = (int)std::min<int64_t>((len + 255) / 256, 2048);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!counter_cleared || attempt > 0) MIH_HIP(hipMemsetAsync(w.sel.p, 0, 2 * sizeof(uint64_t), s));
        hipLaunchKernelGGL(k_threshold, dim3(grid), dim3(256), 0, s, x_dev, len, w.state.p, w.sel.p, (uint32_t)w.cap);
        const int64_t first = std::min<int64_t>(w.expect, w.cap);
        MIH_HIP(hipMemcpyAsync(w.hsel.p, w.sel.p, sizeof(uint64_t) * (2 + 2 * (size_t)first), hipMemcpyDeviceToHost, s));
        MIH_HIP(hipStreamSynchronize(s));
        const int64_t cnt = (int64_t)w.hsel.p[0];
        std::vector<uint64_t> host(w.hsel.p, w.hsel.p + 2 + 2 * (size_t)std::min<int64_t>(cnt, first));
        if (cnt > w.cap) {                     // many exact ties: grow and compact again
            w.cap = cnt + 1024;
            MIH_TRY(w.sel.alloc(2 + 2 * (size_t)w.cap));
            continue;
        }
        if (cnt > first) {                     // more survivors than expected (ties): fetch the rest
            host.resize(2 + 2 * (size_t)cnt);
            MIH_HIP(hipMemcpy(host.data() + 2 + 2 * first, w.sel.p + 2 + 2 * first, sizeof(uint64_t) * 2 * (size_t)(cnt - first), hipMemcpyDeviceToHost));
        }
        std::vector<uint32_t> ord((size_t)cnt);
        for (uint32_t i = 0; i < (uint32_t)cnt; ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return host[2 + 2 * (size_t)a] < host[2 + 2 * (size_t)b]; });
        idx_out.resize((size_t)cnt); val_out.resize((size_t)cnt);
        for (uint32_t i = 0; i < (uint32_t)cnt; ++i) {
            idx_out[i] = (int64_t)host[2 + 2 * (size_t)ord[i]];
            uint64_t bits = host[3 + 2 * (size_t)ord[i]];
            double v; std::memcpy(&v, &bits, sizeof(v));
            val_out[i] = v;
        }
        return MIH_OK;
    }
    set_error("top-k compaction failed");
    return MIH_HIP_ERROR;
}

// the two-pass select; done = false: the prefix is shared by more entries than the gather buffer holds -> 8-bit fallback
static int topk_two_pass(double *x_dev, int64_t len, int64_t k, TopkWork &w, hipStream_t s,
                         std::vector<int64_t> &idx_out, std::vector<double> &val_out, bool zero_in_place, bool &done)
{
    done = false;
    const int gridh = (int)std::min<int64_t>((len + 255) / 256, 512), grid = (int)std::min<int64_t>((len + 255) / 256, 2048);
    hipLaunchKernelGGL(k_hist11, dim3(gridh), dim3(256), 0, s, x_dev, len, 0, w.state.p, w.hist.p);
    hipLaunchKernelGGL(k_pick11, dim3(1), dim3(256), 0, s, w.hist.p, w.state.p, 0, (uint64_t)k, w.sel.p);
    hipLaunchKernelGGL(k_hist11, dim3(gridh), dim3(256), 0, s, x_dev, len, 1, w.state.p, w.hist.p);
    hipLaunchKernelGGL(k_pick11, dim3(1), dim3(256), 0, s, w.hist.p, w.state.p, 1, (uint64_t)k, w.sel.p);
    hipLaunchKernelGGL(k_collect, dim3(grid), dim3(256), 0, s, x_dev, len, w.state.p, w.sel.p, (uint32_t)w.cap);
    const int64_t first = std::min<int64_t>(w.expect, w.cap);
    // (round 6) the exact finish on the device: the survivors come home ranked and ordered (k_topk_finish); the host's own finish
    // below stays for more candidates than that kernel ranks
    static const bool host_finish = probe_env("MENDELIHT_TOPK_HOST_FINISH") != nullptr;        // (measurement build: rounds 1-5's finish)
    if (!host_finish && w.fin.p) {
        hipLaunchKernelGGL(k_topk_finish, dim3(1), dim3(1024), 0, s, w.sel.p, (uint32_t)w.cap, (uint64_t)k, w.state.p, w.fin.p);
        if (zero_in_place) hipLaunchKernelGGL(k_zero_below_dev, dim3(grid), dim3(256), 0, s, x_dev, len, w.fin.p);
        const int64_t fcap = std::min<int64_t>(first, kFinishCap);
        MIH_TRY(readback_words(s, w.flag, w.fin.p, w.hsel.p, 2 + 2 * (size_t)fcap, fcap));
        if (w.hsel.p[0] != ~0ull) {
            const int64_t nsv = (int64_t)w.hsel.p[0];
            std::vector<uint64_t> host(w.hsel.p, w.hsel.p + 2 + 2 * (size_t)std::min<int64_t>(nsv, fcap));
            if (nsv > fcap) {          // more survivors than the landing buffer expected (ties): fetch the rest
                host.resize(2 + 2 * (size_t)nsv);
                MIH_HIP(hipMemcpy(host.data() + 2 + 2 * fcap, w.fin.p + 2 + 2 * fcap, sizeof(uint64_t) * 2 * (size_t)(nsv - fcap), hipMemcpyDeviceToHost));
            }
            idx_out.resize((size_t)nsv); val_out.resize((size_t)nsv);
            for (int64_t t = 0; t < nsv; ++t) {
                idx_out[(size_t)t] = (int64_t)host[2 + 2 * (size_t)t];
                const uint64_t bits = host[3 + 2 * (size_t)t];
                double v; std::memcpy(&v, &bits, sizeof(v));
                val_out[(size_t)t] = v;
            }
            done = true;
            return MIH_OK;
        }
        // (massive ties: the candidates themselves come home, as in rounds 1-5)
    }
    MIH_TRY(readback_words(s, w.flag, w.sel.p, w.hsel.p, 2 + 2 * (size_t)first, first));
    const int64_t cnt = (int64_t)w.hsel.p[0];
    if (cnt > w.cap) return MIH_OK;                        // massive ties: the exact 8-bit select handles any count
    std::vector<uint64_t> host(w.hsel.p, w.hsel.p + 2 + 2 * (size_t)std::min<int64_t>(cnt, first));
    if (cnt > first) {
        host.resize(2 + 2 * (size_t)cnt);
        MIH_HIP(hipMemcpy(host.data() + 2 + 2 * first, w.sel.p + 2 + 2 * first, sizeof(uint64_t) * 2 * (size_t)(cnt - first), hipMemcpyDeviceToHost));
    }
    // the gathered set holds every entry at or above the threshold's 22-bit prefix, hence the k largest of the vector
    uint64_t thr = 0;
    {
        std::vector<uint64_t> keys((size_t)cnt);
        for (int64_t t = 0; t < cnt; ++t) keys[t] = host[3 + 2 * (size_t)t] & 0x7FFFFFFFFFFFFFFFull;
        if (cnt >= k) {
            std::nth_element(keys.begin(), keys.begin() + (k - 1), keys.end(), std::greater<uint64_t>());
            thr = keys[k - 1];                              // |k-th largest| of the whole vector
        }                                                    // fewer than k non-zeros: thr = 0, everything gathered survives
    }
    std::vector<uint32_t> ord;
    ord.reserve((size_t)std::min<int64_t>(cnt, k + 8));
    for (uint32_t t = 0; t < (uint32_t)cnt; ++t) if ((host[3 + 2 * (size_t)t] & 0x7FFFFFFFFFFFFFFFull) >= thr) ord.push_back(t);
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return host[2 + 2 * (size_t)a] < host[2 + 2 * (size_t)b]; });
    idx_out.resize(ord.size()); val_out.resize(ord.size());
    for (size_t i = 0; i < ord.size(); ++i) {
        idx_out[i] = (int64_t)host[2 + 2 * (size_t)ord[i]];
        const uint64_t bits = host[3 + 2 * (size_t)ord[i]];
        double v; std::memcpy(&v, &bits, sizeof(v));
        val_out[i] = v;
    }
    if (zero_in_place) hipLaunchKernelGGL(k_zero_below, dim3(grid), dim3(256), 0, s, x_dev, len, thr);
    done = true;
    return MIH_OK;
}

int topk_project_device(double *x_dev, int64_t len, int64_t k, TopkWork &w, hipStream_t s,
                        std::vector<int64_t> &idx_out, std::vector<double> &val_out, bool zero_in_place)
{
    // utilities.jl:554 DomainError for k<0; partialsort(x, 0) / k>len is a BoundsError
    if (k <= 0 || k > len) { set_error("Attempted to project to sparsity level %lld (vector length %lld)", (long long)k, (long long)len); return MIH_BAD_ARG; }
    if (!w.radix8) {
        bool done = false;
        MIH_TRY(topk_two_pass(x_dev, len, k, w, s, idx_out, val_out, zero_in_place, done));
        if (done) return MIH_OK;
    }
    uint64_t st[4] = {0ull, (uint64_t)k, 0ull, 0ull};
    MIH_HIP(hipMemcpyAsync(w.state.p, st, sizeof(st), hipMemcpyHostToDevice, s));
    int grid = (int)std::min<int64_t>((len + 255) / 256, 512);     // <= 512 blocks: 256 global adds per block at the end
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), 0, s, x_dev, len, shift, w.state.p, w.hist.p);
        hipLaunchKernelGGL(k_pick, dim3(1), dim3(256), 0, s, w.hist.p, w.state.p, shift, w.sel.p);
    }
    return compact_device(x_dev, len, w, s, idx_out, val_out, /*counter_cleared=*/true);
}

int collect_nonzero_device(double *x_dev, int64_t len, TopkWork &w, hipStream_t s,
                           std::vector<int64_t> &idx_out, std::vector<double> &val_out)
{
    uint64_t st[4] = {0ull, 0ull, 0ull, 0ull};       // threshold key 0: nothing is zeroed
    MIH_HIP(hipMemcpyAsync(w.state.p, st, sizeof(st), hipMemcpyHostToDevice, s));
    return compact_device(x_dev, len, w, s, idx_out, val_out);
}

}  // namespace mih

using namespace mih;

extern "C" int mih_project_topk(double *x, int64_t len, int64_t k, int64_t *n_kept)
{
    if (!x || len <= 0) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c == 0) { (void)hipGetLastError(); set_error("no HIP device available"); return MIH_NO_DEVICE; }
    DevBuf<double> d;
    MIH_TRY(d.alloc((size_t)len));
    TopkWork w;
    MIH_TRY(topk_work_init(w, k > 0 ? k : 0));
    hipStream_t s = nullptr;
    MIH_HIP(hipMemcpy(d.p, x, sizeof(double) * (size_t)len, hipMemcpyHostToDevice));
    std::vector<int64_t> idx; std::vector<double> val;
    MIH_TRY(topk_project_device(d.p, len, k, w, s, idx, val));
    MIH_HIP(hipMemcpy(x, d.p, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost));
    if (n_kept) *n_kept = (int64_t)idx.size();
    return MIH_OK;
}
