// group.hip -- project_group_sparse!(y, group, J, k) on the device (src/utilities.jl:613-679).
//
// Reference algorithm: sortperm!(perm, y, by=abs, rev=true) (ties -> ascending index, Base.Order.Perm);
// walk perm accumulating each group's norm over its first k_g members; rank the groups by norm
// (descending, ties -> lower group first); walk perm again keeping a member iff its group's rank <= J
// and it is among the group's first k_g.  Parallel restatement with identical results:
//   1. stable radix sort of indices by |y| bits, descending            (= perm; hand-written, radix_sort_pairs below)
//   2. stable radix sort of perm by group label                        (segments, still |y|-descending)
//   3. one thread per group sums the squares of its first k_g members IN THAT ORDER (same order as
//      the reference's sequential walk, so the norms are bit-identical)
//   4. stable radix sort of groups by norm bits, descending            (= group rank)
//   5. zero every member whose rank > J or whose position in its segment >= k_g.
#include "common.h"

namespace mih {

// ---- stable LSD radix sort of (key, value) pairs, 8 bits a pass ----------------------------------------------------------
// A pass is three launches: per-block digit histograms (a block owns a contiguous tile of kRsTile keys), one exclusive scan
// over the (digit-major, block-minor) histogram -- that order makes the pass stable across blocks -- and the scatter, which
// walks the tile in input order: 256 keys a round, a key's rank among equal digits = equal digits in earlier rounds + in earlier
// waves of the round + in lower lanes of its wave (ballot multi-split).  `descending` sorts by the complemented digit, so equal
// keys keep their input order in both directions (what sortperm / Base.Order.Perm does with ties, utilities.jl:613-679).
constexpr int kRsItems = 8, kRsTile = 256 * kRsItems;
template <typename K>
__global__ void __launch_bounds__(256)
k_rs_hist(const K *__restrict__ key, int64_t len, int shift, int descending, int nblk, uint32_t *__restrict__ hist /* [256][nblk] */)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kRsTile;
    for (int r = 0; r < kRsItems; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        if (i < len) {
            uint32_t d = (uint32_t)(key[i] >> shift) & 255u;
            if (descending) d = 255u - d;
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}
// exclusive scan of `count` 32-bit entries in place (one block; the histogram of a 2^31-key sort is 2^28 entries at most,
// the projection's vectors give ~10^5)
__global__ void __launch_bounds__(1024)
k_rs_scan(uint32_t *__restrict__ a, int64_t count)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < count; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < count ? a[i] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t add = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < count) a[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}
template <typename K>
__global__ void __launch_bounds__(256)
k_rs_scatter(const K *__restrict__ key, const int32_t *__restrict__ val, int64_t len, int shift, int descending, int nblk,
             const uint32_t *__restrict__ hist /* scanned */, K *__restrict__ key_out, int32_t *__restrict__ val_out)
{
    __shared__ uint32_t offs[256];
    __shared__ uint32_t wcnt[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    offs[threadIdx.x] = hist[(int64_t)threadIdx.x * nblk + blockIdx.x];
    #pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kRsTile;
    for (int r = 0; r < kRsItems; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        const bool live = i < len;
        K kk = 0; int32_t vv = 0; uint32_t d = 0;
        if (live) { kk = key[i]; vv = val[i]; d = (uint32_t)(kk >> shift) & 255u; if (descending) d = 255u - d; }
        // lanes of this wave with the same digit
        uint64_t same = __ballot(live);
        #pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (live && rank == 0) wcnt[wave][d] = (uint32_t)__popcll(same);      // the lowest lane of the digit publishes the count
        __syncthreads();
        if (live) {
            uint32_t pos = offs[d] + rank;
            for (int w = 0; w < wave; ++w) pos += wcnt[w][d];
            key_out[pos] = kk; val_out[pos] = vv;
        }
        __syncthreads();
        offs[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        #pragma unroll
        for (int w = 0; w < 4; ++w) wcnt[w][threadIdx.x] = 0;
        __syncthreads();
    }
}
// sorts (key, val) by the low `bits` bits of key; the result is in (key, val) again (bits rounded up to whole bytes, an even
// number of passes by construction of the callers: 64 or 32 bits).  key2 / val2 / hist: scratch of len, len, 256 * nblk entries.
template <typename K>
static void radix_sort_pairs(K *key, K *key2, int32_t *val, int32_t *val2, int64_t len, int bits, int descending,
                             uint32_t *hist, hipStream_t s)
{
    const int nblk = (int)((len + kRsTile - 1) / kRsTile);
    K *ka = key, *kb = key2; int32_t *va = val, *vb = val2;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL((k_rs_hist<K>), dim3(nblk), dim3(256), 0, s, ka, len, shift, descending, nblk, hist);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, s, hist, (int64_t)256 * nblk);
        hipLaunchKernelGGL((k_rs_scatter<K>), dim3(nblk), dim3(256), 0, s, ka, va, len, shift, descending, nblk, hist, kb, vb);
        std::swap(ka, kb); std::swap(va, vb);
    }
}
static size_t radix_sort_hist_entries(int64_t len) { return (size_t)256 * (size_t)((len + kRsTile - 1) / kRsTile); }


__global__ void k_grp_keys(const double *__restrict__ y, int64_t len, uint64_t *__restrict__ key, int32_t *__restrict__ idx)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= len) return;
    key[i] = (uint64_t)__double_as_longlong(y[i]) & 0x7FFFFFFFFFFFFFFFull;
    idx[i] = (int32_t)i;
}
__global__ void k_grp_labels(const int32_t *__restrict__ perm, const int64_t *__restrict__ group, int64_t len,
                             uint32_t *__restrict__ lab)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < len) lab[i] = (uint32_t)(group[perm[i]] - 1);
}
// after the sort by label: seg_start[g] = first position of group g, seg_start[G] = len
__global__ void k_grp_bounds(const uint32_t *__restrict__ lab_sorted, int64_t len, int64_t G, int64_t *__restrict__ seg_start)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= len) return;
    uint32_t g = lab_sorted[i];
    uint32_t prev = (i == 0) ? 0xFFFFFFFFu : lab_sorted[i - 1];
    if (i == 0) { for (uint32_t t = 0; t <= g; ++t) seg_start[t] = 0; }
    else if (g != prev) { for (uint32_t t = prev + 1; t <= g; ++t) seg_start[t] = i; }
    if (i == len - 1) { for (int64_t t = (int64_t)g + 1; t <= G; ++t) seg_start[t] = len; }
}
__global__ void k_grp_norms(const double *__restrict__ y, const int32_t *__restrict__ member, const int64_t *__restrict__ seg_start,
                            int64_t G, const int64_t *__restrict__ k, int k_is_vector, uint64_t *__restrict__ norm_key,
                            int32_t *__restrict__ gid)
{
#pragma clang fp contract(off)       // group_norm[n] + y[j]^2 (utilities.jl:626) rounds the square, then the sum.  __dmul_rn / __dadd_rn are plain
                                     // `*` and `+` in HIP, and the compiler fused them: 2.2^2 + 1.8^2 + 1.3^2 came out as 9.77 instead of
                                     // 9.770000000000001, tied with another group's norm and ranked behind it (seed 9079 of tools/fuzz_parity.py)
    int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= G) return;
    int64_t a = seg_start[g], b = seg_start[g + 1], kg = k_is_vector ? k[g] : k[0];
    double nrm = 0.0;
    for (int64_t t = a; t < b && t - a < kg; ++t) { const double v = y[member[t]]; const double sq = v * v; nrm = nrm + sq; }
    norm_key[g] = (uint64_t)__double_as_longlong(nrm);     // nrm >= 0: bit order == numeric order
    gid[g] = (int32_t)g;
}
__global__ void k_grp_rank(const int32_t *__restrict__ gid_sorted, int64_t G, int32_t *__restrict__ rank)
{
    int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r < G) rank[gid_sorted[r]] = (int32_t)(r + 1);
}
__global__ void k_grp_apply(double *__restrict__ y, const int32_t *__restrict__ member, const uint32_t *__restrict__ lab_sorted,
                            const int64_t *__restrict__ seg_start, const int32_t *__restrict__ rank, int64_t len, int64_t J,
                            const int64_t *__restrict__ k, int k_is_vector)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= len) return;
    uint32_t g = lab_sorted[t];
    int64_t kg = k_is_vector ? k[g] : k[0];
    if (rank[g] > J || t - seg_start[g] >= kg) y[member[t]] = 0.0;
}

// y_dev (len doubles) projected in place; group_dev labels 1..G; k_dev: 1 or G entries
int group_project_device(double *y_dev, const int64_t *group_dev, int64_t len, int64_t G, int64_t J,
                         const int64_t *k_dev, int k_is_vector, hipStream_t s)
{
    if (len >= (1ll << 31)) { set_error("group projection supports len < 2^31"); return MIH_BAD_DIM; }
    // all temporaries out of one block (Arena): ~14 allocations and frees per projection otherwise, every IHT step
    Arena arena;
    const size_t hist_n = std::max(radix_sort_hist_entries(len), radix_sort_hist_entries(G));
    MIH_TRY(arena.reserve((size_t)len * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 4) + (size_t)(G + 1) * (8 + 8 + 8 + 4 + 4 + 4 + 4) + hist_n * 4 + 32 * 256, 0));
    ArenaScope in_arena(&arena);
    DevBuf<uint64_t> key, key2, nkey, nkey2;
    DevBuf<int32_t> idx, perm, member, member2, gid, gid2, rank;
    DevBuf<uint32_t> lab, lab2, hist;
    DevBuf<int64_t> seg;
    MIH_TRY(key.alloc(len)); MIH_TRY(key2.alloc(len)); MIH_TRY(idx.alloc(len)); MIH_TRY(perm.alloc(len));
    MIH_TRY(member.alloc(len)); MIH_TRY(member2.alloc(len)); MIH_TRY(lab.alloc(len)); MIH_TRY(lab2.alloc(len)); MIH_TRY(seg.alloc(G + 1));
    MIH_TRY(nkey.alloc(G)); MIH_TRY(nkey2.alloc(G)); MIH_TRY(gid.alloc(G)); MIH_TRY(gid2.alloc(G)); MIH_TRY(rank.alloc(G));
    MIH_TRY(hist.alloc(hist_n));
    unsigned nb = (unsigned)((len + 255) / 256), gb = (unsigned)((G + 255) / 256);
    hipLaunchKernelGGL(k_grp_keys, dim3(nb), dim3(256), 0, s, y_dev, len, key.p, idx.p);
    radix_sort_pairs<uint64_t>(key.p, key2.p, idx.p, perm.p, len, 64, 1, hist.p, s);            // idx = sortperm(|y|, rev = true)
    hipLaunchKernelGGL(k_grp_labels, dim3(nb), dim3(256), 0, s, idx.p, group_dev, len, lab.p);
    MIH_HIP(hipMemcpyAsync(member.p, idx.p, sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToDevice, s));
    radix_sort_pairs<uint32_t>(lab.p, lab2.p, member.p, member2.p, len, 32, 0, hist.p, s);       // segments by group, |y|-descending inside
    hipLaunchKernelGGL(k_grp_bounds, dim3(nb), dim3(256), 0, s, lab.p, len, G, seg.p);
    hipLaunchKernelGGL(k_grp_norms, dim3(gb), dim3(256), 0, s, y_dev, member.p, seg.p, G, k_dev, k_is_vector, nkey.p, gid.p);
    radix_sort_pairs<uint64_t>(nkey.p, nkey2.p, gid.p, gid2.p, G, 64, 1, hist.p, s);             // groups by norm, descending
    hipLaunchKernelGGL(k_grp_rank, dim3(gb), dim3(256), 0, s, gid.p, G, rank.p);
    hipLaunchKernelGGL(k_grp_apply, dim3(nb), dim3(256), 0, s, y_dev, member.p, lab.p, seg.p, rank.p, len, J, k_dev, k_is_vector);
    MIH_HIP(hipGetLastError());
    MIH_HIP(hipStreamSynchronize(s));     // temporaries are released on return
    return MIH_OK;
}

}  // namespace mih

using namespace mih;

extern "C" int mih_project_group_sparse(double *y, const int64_t *group, int64_t len, int64_t J,
                                        const int64_t *k, int k_is_vector)
{
    if (!y || !group || !k || len <= 0 || J < 0) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c == 0) { (void)hipGetLastError(); set_error("no HIP device available"); return MIH_NO_DEVICE; }
    int64_t G = 0;
    for (int64_t i = 0; i < len; ++i) {
        if (group[i] < 1) { set_error("group labels must be 1..G"); return MIH_BAD_ARG; }
        if (group[i] > G) G = group[i];
    }
    for (int64_t g = 0; g < (k_is_vector ? G : 1); ++g)
        if (k[g] < 0) { set_error("project_group_sparse!: the number of predictors per group must be nonnegative"); return MIH_BAD_ARG; }
    DevBuf<double> dy; DevBuf<int64_t> dg, dk;
    MIH_TRY(dy.alloc(len)); MIH_TRY(dg.alloc(len)); MIH_TRY(dk.alloc(k_is_vector ? G : 1));
    MIH_HIP(hipMemcpy(dy.p, y, sizeof(double) * len, hipMemcpyHostToDevice));
    MIH_HIP(hipMemcpy(dg.p, group, sizeof(int64_t) * len, hipMemcpyHostToDevice));
    MIH_HIP(hipMemcpy(dk.p, k, sizeof(int64_t) * (k_is_vector ? G : 1), hipMemcpyHostToDevice));
    MIH_TRY(group_project_device(dy.p, dg.p, len, G, J, dk.p, k_is_vector, nullptr));
    MIH_HIP(hipMemcpy(y, dy.p, sizeof(double) * len, hipMemcpyDeviceToHost));
    return MIH_OK;
}
