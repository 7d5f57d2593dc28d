// common.h -- internal declarations shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <iterator>
#include <memory>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <functional>
#include <vector>
#include "../../include/mendeliht_hip.h"

namespace mih {

void set_error(const char *fmt, ...);
int  hip_fail(hipError_t e, const char *what, const char *file, int line);

// A/B switches and tuning knobs of the MEASUREMENT build (-DMIH_PROBES: libmendeliht_hip_probes.so, used by tools/ and by the
// "this switch changes nothing" tests).  The release library does not read them: its behaviour is fixed by its arguments.
inline const char *probe_env(const char *name)
{
#ifdef MIH_PROBES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

#define MIH_HIP(expr)                                                         \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) return ::mih::hip_fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)

#define MIH_TRY(expr)                      \
    do {                                   \
        int rc_ = (expr);                  \
        if (rc_ != MIH_OK) return rc_;     \
    } while (0)

// Rows per r-tile chunk: one wave-load of 64 dwords = 64 lanes x 16 genotypes.
constexpr int kChunkRows = 1024;

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// A large matrix keeps a reserve of device memory for the fits that will run on it (DevPool): the lock-step lanes' workspaces
// and the IHTVariables' blocks are cut out of it, so a fit never calls hipMalloc / hipFree.  Reasons: (1) on this driver the
// first hipMalloc that lands in a large never-used block of VRAM stalls ~2.9 s while the block is cleared -- in 7 of 16 fresh
// processes that was the creation of an IHTVariable inside the first cv_iht (6 s instead of 3 s; an allocate-and-release
// warm-up at matrix creation cut it to 1 in 16, holding the memory removes it); (2) every hipFree waits for the device.
// First fit with coalescing; 2 MB granularity; requests it cannot serve fall back to hipMalloc.
struct DevPool {
    char *base = nullptr; size_t bytes = 0;
    std::mutex mu;
    std::map<size_t, size_t> free_blocks;      // offset -> size
    std::map<size_t, size_t> used;             // offset -> size
    static constexpr size_t kGran = 2ull << 20;
    ~DevPool() { if (base) (void)hipFree(base); }
    bool init(size_t total)
    {
        total = (total + kGran - 1) / kGran * kGran;
        if (hipMalloc((void **)&base, total) != hipSuccess) { (void)hipGetLastError(); base = nullptr; return false; }
        bytes = total; free_blocks[0] = total;
        return true;
    }
    void *take(size_t want)
    {
        want = (want + kGran - 1) / kGran * kGran;
        std::lock_guard<std::mutex> g(mu);
        for (auto it = free_blocks.begin(); it != free_blocks.end(); ++it) {
            if (it->second < want) continue;
            const size_t off = it->first, sz = it->second;
            free_blocks.erase(it);
            if (sz > want) free_blocks[off + want] = sz - want;
            used[off] = want;
            return base + off;
        }
        return nullptr;
    }
    bool owns(const void *q) const { return base && (const char *)q >= base && (const char *)q < base + bytes; }
    void give_back(void *q)
    {
        std::lock_guard<std::mutex> g(mu);
        const size_t off = (size_t)((char *)q - base);
        auto u = used.find(off);
        if (u == used.end()) return;
        size_t o = off, sz = u->second;
        used.erase(u);
        auto nx = free_blocks.lower_bound(o);
        if (nx != free_blocks.end() && o + sz == nx->first) { sz += nx->second; nx = free_blocks.erase(nx); }
        if (nx != free_blocks.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == o) { o = pv->first; sz += pv->second; free_blocks.erase(pv); } }
        free_blocks[o] = sz;
    }
};
inline DevPool *&current_pool() { static thread_local DevPool *q = nullptr; return q; }
struct PoolScope {          // (a lane opens ONE for its whole thread, before any coroutine starts: not a coop_blocked scope)
    DevPool *prev;
    explicit PoolScope(DevPool *q) : prev(current_pool()) { current_pool() = q; }
    ~PoolScope() { current_pool() = prev; }
};
// device memory from the thread's current pool if it has room (and the request is worth a 2 MB granule), else hipMalloc
inline hipError_t dev_malloc(void **q, size_t bytes, DevPool **from)
{
    *from = nullptr;
    if (DevPool *pl = current_pool()) {
        if (bytes >= (1ull << 20)) { if (void *r = pl->take(bytes)) { *q = r; *from = pl; return hipSuccess; } }
    }
    return hipMalloc(q, bytes);
}

// An IHTVariable owns ~26 device and ~5 pinned buffers.  One hipMalloc of a few MB costs ~100 us and every hipFree ~150 us
// (it also waits for whatever the device is running): 6 ms per IHTVariable, 14 % of a GPU's 13-fit share of a cross-validation
// (tools/pin_alloc_time.hip).  While an Arena is the thread's current one (ArenaScope), DevBuf / PinBuf allocations are carved
// out of its single block (256-byte aligned) and do not own their memory; when the block is exhausted, or outside a scope
// (later growth), they fall back to their own hipMalloc / hipHostMalloc.
struct Arena {
    char *dev = nullptr, *pin = nullptr;
    DevPool *dev_pool = nullptr;
    size_t dev_bytes = 0, pin_bytes = 0, dev_off = 0, pin_off = 0;
    Arena() = default;
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    ~Arena() { release_dev(); if (pin) (void)hipHostFree(pin); }
    void release_dev() { if (dev) { if (dev_pool) dev_pool->give_back(dev); else (void)hipFree(dev); dev = nullptr; } }
    int reserve(size_t device_bytes, size_t pinned_bytes)
    {
        static const bool off = probe_env("MENDELIHT_NO_ARENA") != nullptr;       // A/B: every buffer its own allocation
        if (off) return MIH_OK;
        if (dev_malloc((void **)&dev, device_bytes, &dev_pool) != hipSuccess) { (void)hipGetLastError(); dev = nullptr; device_bytes = 0; }   // fall back to single buffers
        if (hipHostMalloc((void **)&pin, pinned_bytes, hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); pin = nullptr; pinned_bytes = 0; }
        dev_bytes = device_bytes; pin_bytes = pinned_bytes; dev_off = pin_off = 0;
        return MIH_OK;
    }
    void *take_dev(size_t bytes)
    {
        const size_t o = (dev_off + 255) & ~(size_t)255;
        if (!dev || o + bytes > dev_bytes) return nullptr;
        dev_off = o + bytes;
        return dev + o;
    }
    void *take_pin(size_t bytes)
    {
        const size_t o = (pin_off + 255) & ~(size_t)255;
        if (!pin || o + bytes > pin_bytes) return nullptr;
        pin_off = o + bytes;
        return pin + o;
    }
};
inline Arena *&current_arena() { static thread_local Arena *a = nullptr; return a; }
// Cooperative waits.  In a lock-step round the fits of a lane are coroutines of ONE host thread (CoopSched; LaneSched of fit_common.h, driven by fit_lockstep.hip): a fit that
// reaches a readback does not spin -- it yields, and the thread queues the next fit's kernels meanwhile.  Allocation scopes are
// thread-local state living on a coroutine's stack, so nothing may yield while one is open (coop_blocked).
struct CoopSched { virtual void yield() = 0; virtual ~CoopSched() {} };
inline CoopSched *&current_coop() { static thread_local CoopSched *c = nullptr; return c; }
inline int &coop_blocked() { static thread_local int d = 0; return d; }
inline bool coop_can_yield() { return current_coop() != nullptr && coop_blocked() == 0; }
// hipStreamSynchronize that lets the lane's other fits run while this one's stream drains
int stream_sync_coop(hipStream_t s);
struct ArenaScope {
    Arena *prev;
    explicit ArenaScope(Arena *a) : prev(current_arena()) { current_arena() = a; ++coop_blocked(); }
    ~ArenaScope() { current_arena() = prev; --coop_blocked(); }
};

// RAII device buffer
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool own = true;           // false: carved out of an Arena (which outlives the buffer's owner)
    DevPool *pool = nullptr;   // own && pool: a block of the matrix's DevPool
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p && own) { if (pool) pool->give_back(p); else (void)hipFree(p); } p = nullptr; n = 0; own = true; pool = nullptr; }
    void attach(T *ptr, size_t count) { release(); p = ptr; n = count; own = false; }     // someone else's memory (read-only sharing)
    int alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        if (Arena *a = current_arena()) {
            if (void *q = a->take_dev(count * sizeof(T))) { p = static_cast<T *>(q); n = count; own = false; return MIH_OK; }
        }
        hipError_t e = dev_malloc((void **)&p, count * sizeof(T), &pool);
        if (e != hipSuccess) { set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e)); p = nullptr; return e == hipErrorOutOfMemory ? MIH_OOM : MIH_HIP_ERROR; }
        n = count;
        return MIH_OK;
    }
};

// pinned host scratch for the small per-iteration readbacks (a device-to-host copy into pageable memory is
// staged by the runtime and costs ~15 us more per host synchronisation)
template <typename T>
struct PinBuf {
    T *p = nullptr;
    size_t n = 0;
    bool own = true;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { if (p && own) (void)hipHostFree(p); }
    // coherent = true: fine-grained host memory a kernel may write while the host polls it (SpinFlag landing areas)
    int alloc(size_t count, bool coherent = false) {
        if (p && own) (void)hipHostFree(p);
        p = nullptr; n = 0; own = true;
        if (count == 0) count = 1;
        if (Arena *a = current_arena()) {            // arena memory is coherent: good for either kind
            if (void *q = a->take_pin(count * sizeof(T))) { p = static_cast<T *>(q); n = count; own = false; return MIH_OK; }
        }
        if (hipHostMalloc((void **)&p, count * sizeof(T), coherent ? hipHostMallocCoherent : hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc(%zu bytes) failed", count * sizeof(T)); (void)hipGetLastError(); p = nullptr; return MIH_OOM; }
        n = count;
        return MIH_OK;
    }
};

// Small results come home without a stream synchronisation: the LAST kernel of a chain (k_publish, one block) copies them
// into pinned host memory itself and then stores a sequence number there (system-scope release); the host spins on that
// word.  A device-to-host copy followed by hipStreamSynchronize costs one more queued operation and 25-30 us of wake-up
// latency per readback, three to five times per IHT iteration.  After `spin_us` without the flag the host falls back to
// hipStreamSynchronize (a fused pass of tens of ms is in front of the chain, or something failed: the error surfaces there).
// MENDELIHT_NO_SPIN=1: always copy + hipStreamSynchronize.
struct SpinFlag {
    PinBuf<uint64_t> word;          // [0] = sequence number of the last published readback
    uint64_t seq = 0;
};
// dst_host[0 .. words) <- src_dev[0 .. words) (64-bit words); pairs_first >= 0: src_dev = {count, pad, pairs...} and only
// 2 + 2 * min(count, pairs_first) words travel (the projection's survivor list).  Returns after the data has landed.
int readback_words(hipStream_t s, SpinFlag &f, const uint64_t *src_dev, uint64_t *dst_host, size_t words, int64_t pairs_first = -1);
// The same in two halves, for a chain whose last kernel publishes by itself (publish_block): spin_begin hands out the sequence
// number that kernel must store (0: polling is off, use a copy + hipStreamSynchronize), spin_wait returns once it is there.
uint64_t spin_begin(SpinFlag &f);
int spin_wait(hipStream_t s, SpinFlag &f, uint64_t seq);
#if defined(__HIPCC__)
// whole block: words of device memory -> pinned host memory, then the sequence number (system-scope release)
__device__ __forceinline__ void publish_block(const uint64_t *src, uint64_t *dst_host, uint64_t words, uint64_t *flag_host, uint64_t seq)
{
    for (uint64_t i = threadIdx.x; i < words; i += blockDim.x)
        dst_host[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // past the CU's vector cache
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

// Small host lists (support indices, coefficients, cache slots) reach the device without copy operations: the host writes
// them into a slot of a ring of pinned memory and the consuming kernel reads that slot itself, once per entry.  A copy from a
// pageable std::vector costs 10-20 us of host time and a queued operation each, four to seven times per IHT iteration.
// A slot is reused after kSlots - 1 further puts; put() synchronises the stream first unless the owner has reported a
// synchronisation (synced(): every readback is one) since the slot was handed out.
struct HostStage {
    static constexpr int kSlots = 8;
    PinBuf<uint64_t> ring;
    size_t slot_words = 0;
    int next = 0, since_sync = 0;
    int init(size_t words_per_slot) { slot_words = words_per_slot; next = 0; since_sync = 0; return ring.alloc(words_per_slot * kSlots, true); }
    // nullptr in *out: does not fit a slot (the caller copies the old way)
    int put(hipStream_t s, const void *a, size_t bytes_a, const void *b, size_t bytes_b, const uint64_t **out);
    void synced() { since_sync = 0; }
};
// dst_a[0 .. words_a) <- src[0 .. words_a), dst_b[0 .. words_b) <- src[words_a ..): one launch instead of two copies
void stage_to_device(hipStream_t s, const uint64_t *src_pinned, uint64_t *dst_a, size_t words_a, uint64_t *dst_b, size_t words_b);

}  // namespace mih

namespace mih {
// Measurement hook of a matrix handle (mih_profile_*): when enabled, every launch of the dominant X'r kernel on this matrix is
// bracketed by HIP events on the stream it runs on and recorded with its kernel name and residual count; the lock-step drivers
// also count what they did (lanes, slots, hand-overs, shared initial scores).  Lives behind the handle, not in the process.
struct PassRecord { hipEvent_t e0 = nullptr, e1 = nullptr; int residuals = 0, operands = 0, stream_tag = 0; char kernel[48] = {0}; };
// exchanges of a column-sharded fit (mih_profile_exchange): [0] all-reduce of n + 1 doubles (X_S g_S with |df_S|^2 riding along),
// [1] all-reduce of n doubles (X_S b_S), [2] all-gather of the projection's candidates, [3] scalar exchanges on the host
struct ExchRecord { hipEvent_t e0 = nullptr, e1 = nullptr; int kind = 0; };
struct Profile {
    std::mutex mu;
    std::atomic<bool> on{false};                 // read by the lanes' threads without the mutex
    hipEvent_t origin = nullptr;                 // recorded when profiling was switched on: start offsets are relative to it
    std::vector<PassRecord> open;                // launches whose events have not been read yet
    std::vector<mih_pass_record> done;
    int64_t counters[MIH_PROFILE_NCOUNTERS] = {0};
    std::vector<ExchRecord> xopen;               // collectives queued on a fit's stream, events not read yet
    double xms[4] = {0.0, 0.0, 0.0, 0.0}; int64_t xcount[4] = {0, 0, 0, 0};
    void exch_host(int kind, double ms) { if (!on) return; std::lock_guard<std::mutex> g(mu); xms[kind] += ms; ++xcount[kind]; }
    void count(int which, int64_t add) { if (!on) return; std::lock_guard<std::mutex> g(mu); counters[which] += add; }
    void count_max(int which, int64_t v) { if (!on) return; std::lock_guard<std::mutex> g(mu); if (v > counters[which]) counters[which] = v; }
    void drain();                                // synchronise the open records into `done`
    ~Profile();
};
}  // namespace mih

// Device-resident design matrix.
//
// kind 0 (SnpLinAlg): 2-bit dosage codes in a TILE-MAJOR layout built for the matrix cores.
// A tile is 32 SNP columns x 128 rows = 1 KB = one wave-load of 16 B per lane:
//     X[(cg * nbp + bp) * 64 + lane] : uint4,   lane = 32*h + m
// where cg = column / 32, m = column % 32, bp = row / 128, and the lane's four dwords are
// {e=0,u=0}, {e=0,u=1}, {e=1,u=0}, {e=1,u=1} covering rows 128*bp + 64*e + 32*h + 16*u + (0..15),
// two bits per row, low bits first.  With this layout lane (m, h) of a wave holds exactly the
// A-operand fragment (row m, K-half h) of v_mfma_scale_f32_32x32x64_f8f6f4 for both 64-row blocks.
// The code is the DOSAGE itself (00->0, 01->1, 10->2; PLINK's code is remapped once at upload):
// placed in the low bits of a nibble it is the FP4 (e2m1) number dosage/2.  Missing entries are
// stored as 0 with their row numbers in a per-column CSR side list; pad rows/columns are 0.
struct mih_mat {
    int       kind = 0;            // 0 snp, 1 dense
    int       device = 0;
    int64_t   n = 0, p = 0;
    int       center = 1, scale = 1, impute = 1;
    int64_t   ncg = 0;             // column groups of 32
    int64_t   nbp = 0;             // row block pairs of 128
    int64_t   n_pad = 0;           // nbp * 128
    uint32_t *X = nullptr;         // ncg * nbp * 64 * 4 dwords
    double   *mu = nullptr, *sinv = nullptr;   // p
    int64_t  *miss_ptr = nullptr;  // p+1
    int32_t  *miss_row = nullptr;  // total_missing
    int64_t   total_missing = 0;
    double   *D = nullptr;         // dense n x p (Float64 storage)
    float    *Df = nullptr;        // dense n x p (Float32 storage: `x::Matrix{Float32}`; arithmetic stays f64)
    hipStream_t stream = nullptr;  // for the stand-alone linear-algebra entry points
    mih::DevPool *pool = nullptr;  // reserve for the fits that run on this matrix (large 2-bit matrices only)
    std::shared_ptr<mih::DevPool> pool_owner;      // a session keeps a reference: the reserve outlives a matrix destroyed first
    std::shared_ptr<mih::Profile> prof = std::make_shared<mih::Profile>();    // measurement hook (mih_profile_*), off by default
    // Streams for the small per-fit kernel chains of the lock-step drivers (worker_stream below): created on first use, kept for
    // the life of the matrix -- hipStreamCreate costs ~4 ms, a fresh set per cross-validation would cost more than it saves.
    mutable std::mutex ws_mu;
    mutable std::vector<hipStream_t> worker_streams;
};
namespace mih {
constexpr int kWorkerStreamsPerLane = 4;       // the runtime maps streams onto a handful of hardware queues anyway
// stream i of the matrix's worker set (i < 2 * kWorkerStreamsPerLane); nullptr if it cannot be created
hipStream_t worker_stream(const mih_mat *h, int i);
}

namespace mih {

// ---- X'r ---------------------------------------------------------------------
// index (in dwords) of the 16 rows 16*t .. 16*t+15 of column j inside the tile-major layout
__host__ __device__ inline int64_t xword(int64_t nbp, int64_t j, int64_t t)
{
    int64_t cg = j >> 5, blk = t >> 2, bp = blk >> 1;
    int m = (int)(j & 31), h = (int)((t >> 1) & 1), u = (int)(t & 1), e = (int)(blk & 1);
    return (((cg * nbp + bp) * 64 + (h * 32 + m)) << 2) + (e << 1) + u;
}

// How a residual is written as digit planes (B-operand columns of the block-scaled MFMA).
//   base 49: FP6 (e2m3) digits d/8 with d in {-32..-18 even step, -16..16, 18..32 even}  (a complete residue system mod 49)
//   base 13: FP4 (e2m1) digits d/2 with d in {-8,-6,-4..4,6,8}
//   base  4: FP4 digits d/2 with d in {-2,-1,0,1}
struct DigitMode {
    int base;        // 49, 13 or 4
    int ndig;        // digits per residual
    int per_op;      // residuals per 32-column B operand
    int slots;       // columns per residual (ndig <= slots, per_op * slots <= 32)
    int ebits;       // the residual is scaled to max|r| * 2^e < 2^(ebits+1)
    int rows_log2;   // a row slice holds at most 2^rows_log2 rows (f32 accumulators stay exact)
    int lay16 = 0;   // FP6 planes stored as the B fragments of the 16x16x128 MFMA (two 16-column images per 128-row block)
    int flat = 0;    // (round 4) the residuals of a pass occupy digit columns 10 j .. 10 j + 9 of the pass's operands back to back,
                     // across operand boundaries: 19 ten-digit residuals in the 192 columns of six operands instead of 18
    int nres = 0;    // flat: residuals in this pass
    // device-resident steps (fit_state.h): the digit kernel and the single-fit pass run only while *gate == gate_val -- a chain queued
    // ahead of the host's knowledge (a step that turned out to need backtracking, a fit that converged) then does nothing
    const int32_t *gate = nullptr; int32_t gate_val = 0;
};
// flat packing: which pass a residual rides and where the pass's operands start (k_digits)
constexpr int kMaxFlatPasses = 16;
struct FlatPasses { int npass; int u0[kMaxFlatPasses + 1]; int t0[kMaxFlatPasses + 1]; };
// How the X'r passes of a workspace run.  Fixed when the workspace is built, from the CALL's arguments (mih_fit_params::
// xtv_digits, the digits argument of mih_xtv_batched_fmt / mih_bench_xtv): there is no process-wide kernel or format selector.
// Everything but `digits` is a knob of the measurement build (mih_probe_*; constant in the release library).
struct XtvTune {
    int digits = 0;          // residual format id (include/mendeliht_hip.h), 0 = library default
    int variant = -1;        // >= 0: a per-wave-load single-operand shape (round-1 kernels)
    int multi_variant = 0;   // launch-shape / probe id of the LDS-shared and ring kernels
    int max_nr = 4;          // B operands fused per pass of the register-staged kernels (1, 2 or 4)
    int max_ops = 6;         // B operands per pass of the 16x16x128 ring kernel: 6 = 18 residuals, 19 with the digit columns packed flat (round 4) (254 VGPRs, ring depth 3).  Round 3,
                             // after the LDS fix: 18 residuals 38.0 ms = 2.11 ms each against 15 in 33.9 ms = 2.26 ms each; the
                             // 100-fit cv_iht 2.82-2.84 s against 2.95-3.03 s with 5 (74 instead of 88 fused passes, same box)
    int slices = 0;          // row slices, 0 = auto_splits
    bool half = true;        // leave out the empty second fragment of a pass's last operand
};
// the tuning of a call: digits from the caller, the rest defaults (release) or the probe knobs (measurement build)
XtvTune xtv_tune(int digits);
inline XtvTune xtv_tune(const mih_fit_params *prm) { return xtv_tune(prm ? prm->xtv_digits : 0); }
bool xtv_digits_valid(int digits);

// Device-resident steps (fit_state.h): the finalize kernel of a single-residual pass also leaves df on the support of the current iterate
// and k_xv_coef's coefficients of X_S df_S (iht_stepsize!) -- extra workgroups that redo the finalize arithmetic of those columns.
struct XtvSupportHook {
    const int32_t *cur = nullptr;                 // which of the two lists is the current iterate's (nullptr: no hook)
    const int64_t *idx[2] = {nullptr, nullptr};
    const int64_t *cnt[2] = {nullptr, nullptr};   // the lists' lengths, in device memory
    double *gval = nullptr, *A = nullptr, *B = nullptr;
    int blocks = 0;                               // extra workgroups of 256 entries
};
// ... and the digit kernel finishes the residual statistics (the second stage of k_r_stats over its 64 block partials) and Z'r (the
// second stage of k_zt_r over 128 block partials per covariate) that k_res_stats left as partials
struct XtvStatsHook {
    const double *spart = nullptr;                // [64][2] max |r|, sum r per block (nullptr: no hook)
    const double *zpart = nullptr; double *df2 = nullptr; int q = 0, zblocks = 0;
    int ebits = 0;
};
// The fused passes of a matrix's lock-step lanes run ONE AFTER THE OTHER (round 6): every pass waits for the pass queued before it,
// whichever lane queued it.  Two passes dispatched side by side share the CUs and end together, the lanes fall into step, and each
// round then ends with a phase of small per-fit kernels that no pass overlaps; in single file lane A's chains run under lane B's
// pass and the matrix pipe never waits (tools/trace_cv_rounds.sh: 85 -> 76 ms per pair of passes).
struct PassOrder { std::mutex mu; hipEvent_t last = nullptr; };
struct XtvWork {            // scratch for one in-flight X'r
    DevBuf<uint32_t> digits;   // ops * nblk * 64 lanes * 4 dwords (+ 2 dwords, stored behind, for FP6) : digit planes of r (B operands)
    DevBuf<double>   partial;  // splits * rhs * ncg*32 raw dots
    DevBuf<double>   scal;     // rhs * 4 : {max|r|, 2^-e, sum r, 2^e}
    DevBuf<unsigned> stat_done; // rhs : blocks of k_r_stats that have delivered their partial (zero between launches)
    double peels_counted = 0.0; // ... of whose guards' running counts the measurement hook has taken this much (xtv_count_peels)
    DevBuf<double>   peel;     // rhs * kPeelStride : the rows of each residual that ride the f64 side channel instead of the fixed point (peel.h)
    int m_cap = 0, splits_cap = 0;
    DigitMode dm = {13, 16, 2, 16, 56, 20};   // fixed at init (tune.digits and the matrix height)
    int ops_cap = 0;           // B operands the buffers hold
    size_t rhs_cap = 0;        // residuals the statistics / partial buffers hold
    XtvTune tune;              // fixed at init
    int stream_tag = 0;        // which lock-step lane launches on this workspace (profile records)
    const int32_t *gate = nullptr; int32_t gate_val = 0;   // see DigitMode::gate; set by the caller around one xtv_device call
    bool stats_done = false;   // ... whose residual statistics (scal) the caller has computed already (k_res_stats, fit_state.h)
    // xtv_digits = -1 (auto): the lock-step drivers score a residual whose max |r| / rms(r) is small in the 43-bit format, the others
    // in the 54-bit one -- per RESIDUAL (a fit's bits never depend on its company); use_alt selects the format of ONE xtv_device call
    DigitMode dm_alt = {49, 8, 4, 8, 42, 18}; bool has_alt = false, use_alt = false;
    PassOrder *order = nullptr; hipEvent_t pass_done = nullptr;   // a lane's workspace: its passes take their turn (PassOrder); pass_done is this lane's event
    XtvSupportHook hook;       // ... and whose finalize kernel also serves the support of the iterate
    XtvStatsHook shook;        // ... and whose digit kernel finishes those statistics
};
// batched = false: the workspace of a single univariate fit (one residual per pass); true: fused multi-RHS passes
int  xtv_work_init(const mih_mat *h, XtvWork &w, int m, const XtvTune &tune, bool batched = true);
// r_dev: m vectors of length n (column-major n x m) on device; out_dev p x m.
int  xtv_device(const mih_mat *h, XtvWork &w, const double *r_dev, int m, double *out_dev, hipStream_t s);
void xtv_count_peels(const mih_mat *h, XtvWork &w, hipStream_t s);
// residuals of two full fused passes (six operands each by default) in the batched format: how many fits the lock-step drivers keep in flight
int  xtv_lockstep_width(const mih_mat *h, const XtvTune &tune);

// ---- X[:,S] v -----------------------------------------------------------------
struct XvWork {
    DevBuf<double> partial;   // groups * n (dense matrices only)
    DevBuf<double> coefA, coefB;  // per support column: sinv*val, -mu*sinv*val
    DevBuf<double> coefG;         // multi-trait products: per trait and column group, the group's sum of coefB
    int64_t cap = 0; int groups = 0;
    // Column cache (2-bit matrices): in the tile-major layout a column shares every 64 B sector with its tile
    // neighbours, so reading k support columns costs 4-8x their size.  The support changes slowly, so each
    // column is copied ONCE into a contiguous slot (ndw dwords) and X*v reads the slots (LRU replacement).
    DevBuf<uint32_t> cache;   // slots * ndw
    DevBuf<int32_t>  slot_dev;
    DevBuf<int64_t>  fill_dev;
    DevBuf<int64_t>  off_dev;            // k_xv_snp_cached_mt: byte offsets of the support columns' cache slots, padded (kXvPadCols)
    int64_t slots = 0;
    std::unordered_map<int64_t, int32_t> slot_of;
    std::vector<int64_t> col_of;
    std::vector<uint64_t> stamp;
    uint64_t tick = 0;
    std::vector<int64_t> h_offs;
    std::vector<int32_t> h_slots; std::vector<int64_t> h_fills;   // host images of slot_dev / fill_dev of the last call
};
int  xv_work_init(const mih_mat *h, XvWork &w, int64_t max_nnz, int64_t cache_nnz = 0);
size_t xv_work_bytes(const mih_mat *h, int64_t max_nnz, int64_t cache_nnz = 0);
// out[i] = sum_t x[i, idx[t]] * val[t]; idx/val on device; clamp20 applies clamp!(out,-20,20).  idx_host (the same
// indices on the host) enables the column cache.
int  xv_sparse_device(const mih_mat *h, XvWork &w, const int64_t *idx_dev, const double *val_dev,
                      int64_t nnz, double *out_dev, int clamp20, hipStream_t s, const int64_t *idx_host = nullptr,
                      HostStage *st = nullptr, const double *gather_src = nullptr, double *gather_out = nullptr);
// gather_src != nullptr: val_dev is ignored, the coefficients are gather_src[idx[t]] and are also left in gather_out

// m coefficient vectors over the same support (vals_dev[v*nnz + t], out_dev[v*n + i]); no clamp
int  xv_sparse_multi_device(const mih_mat *h, XvWork &w, const int64_t *idx_dev, const double *vals_dev, int64_t nnz, int m,
                            double *out_dev, hipStream_t s, const int64_t *idx_host);

// ---- top-k --------------------------------------------------------------------
struct TopkWork {
    DevBuf<uint32_t> hist;     // 256 bins
    DevBuf<uint64_t> state;    // [0]=prefix, [1]=remaining k, [2]=threshold bits, [3]=count_ge
    PinBuf<uint64_t> hsel;     // pinned landing buffer for the count + the first `expect` pairs
    DevBuf<uint64_t> sel;      // compacted survivors as (index, value bits) pairs; pair 0 holds the count
    DevBuf<uint64_t> fin;      // k_topk_finish: count, threshold key, the survivors ordered by index
    int64_t cap = 0;           // pairs the buffer can hold after the header
    int64_t expect = 0;        // survivors expected by the caller (k + slack): fetched with the count in ONE copy
    bool radix8 = false;       // MENDELIHT_TOPK_RADIX8=1: always the 8 x 8-bit select (the two-pass select's fallback)
    SpinFlag flag;             // the survivor list comes home through k_publish
};
int  topk_work_init(TopkWork &w, int64_t max_keep);
// In-place project_k! on a device vector; returns threshold and survivors (sorted by index) on host.
// zero_in_place = false: the caller only needs the survivor lists (the device vector may be left unprojected)
int  topk_project_device(double *x_dev, int64_t len, int64_t k, TopkWork &w, hipStream_t s,
                         std::vector<int64_t> &idx_out, std::vector<double> &val_out, bool zero_in_place = true);
// every non-zero of a device vector as (index, value), sorted by index
int  collect_nonzero_device(double *x_dev, int64_t len, TopkWork &w, hipStream_t s,
                            std::vector<int64_t> &idx_out, std::vector<double> &val_out);
// project_group_sparse! in place on a device vector (group labels 1..G, k_dev: 1 or G entries)
int  group_project_device(double *y_dev, const int64_t *group_dev, int64_t len, int64_t G, int64_t J,
                          const int64_t *k_dev, int k_is_vector, hipStream_t s);
// debias! (utilities.jl:1014-1020): GLM refit of y on the k support columns (debias.hip); beta_out[k] on the host
// A column shard (round 6): `k` is the shard's own part of the support, k_total the whole support's size and k_off the place of the
// shard's first column in it (the shards' column blocks follow each other in rank order, so the whole support is the shards' lists
// one after the other); the shard decodes its own columns into an n x k_total panel of zeros and `reduce` sums the panels of all
// shards (an exact sum: one non-zero contribution per entry); every shard then runs the SAME refit and gets all k_total coefficients.
struct DebiasShard { int64_t k_total = 0, k_off = 0; std::function<int(double *, int64_t)> reduce; };
int  debias_glm_device(const mih_mat *h, const int64_t *idx_host, int64_t k, const double *y_dev, int dist, int link,
                       double nb_r, double *beta_out, hipStream_t s, const DebiasShard *shard = nullptr);
// ncclAllReduce on the fit's own stream when `c` is the library's communicator (comm.hip); -1 otherwise
int  comm_native_allreduce_on_stream(const mih_comm *c, double *buf_dev, int64_t count, int32_t op, hipStream_t s, int device);
int  comm_native_allgather_on_stream(const mih_comm *c, const double *send_dev, double *recv_dev, int64_t count, hipStream_t s, int device);
bool comm_is_native(const mih_comm *c, int device);       // the library's own RCCL communicator, on that device
// rank_of[fold * npath + ik]: which rank of `world` evaluates that (fold, k) combination (mih_cv_assignment; fit_lockstep.hip)
void cv_assign(const int64_t *path, int64_t npath, int32_t nfolds, int32_t world, std::vector<int32_t> &rank_of);
// initialize_beta! regressions for m response planes (fit.hip); shared by the univariate and multivariate fits
int  init_beta_regress_device(const mih_mat *h, const double *w_dev, const double *Y_dev, int m, double N,
                              const double *Sy_host, double *beta_dev, double *icpt_sum_host,
                              DevBuf<double> &red, DevBuf<double> &scal, hipStream_t s, const XtvTune &tune);

}  // namespace mih
