// group.hip -- project_group_sparse!(y, group, J, k) on the device (src/utilities.jl:613-679).
//
// Reference algorithm: sortperm!(perm, y, by=abs, rev=true) (ties -> ascending index, Base.Order.Perm);
// walk perm accumulating each group's norm over its first k_g members; rank the groups by norm
// (descending, ties -> lower group first); walk perm again keeping a member iff its group's rank <= J
// and it is among the group's first k_g.  Parallel restatement with identical results:
//   1. stable radix sort of indices by |y| bits, descending            (= perm)
//   2. stable radix sort of perm by group label                        (segments, still |y|-descending)
//   3. one thread per group sums the squares of its first k_g members IN THAT ORDER (same order as
//      the reference's sequential walk, so the norms are bit-identical)
//   4. stable radix sort of groups by norm bits, descending            (= group rank)
//   5. zero every member whose rank > J or whose position in its segment >= k_g.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace mih {

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
    int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= G) return;
    int64_t a = seg_start[g], b = seg_start[g + 1], kg = k_is_vector ? k[g] : k[0];
    double nrm = 0.0;
    for (int64_t t = a; t < b && t - a < kg; ++t) { double v = y[member[t]]; nrm = __dadd_rn(nrm, __dmul_rn(v, v)); }   // unfused, as utilities.jl:626
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
    DevBuf<uint64_t> key, key2, nkey, nkey2;
    DevBuf<int32_t> idx, perm, member, gid, gid2, rank;
    DevBuf<uint32_t> lab, lab2;
    DevBuf<int64_t> seg;
    MIH_TRY(key.alloc(len)); MIH_TRY(key2.alloc(len)); MIH_TRY(idx.alloc(len)); MIH_TRY(perm.alloc(len));
    MIH_TRY(member.alloc(len)); MIH_TRY(lab.alloc(len)); MIH_TRY(lab2.alloc(len)); MIH_TRY(seg.alloc(G + 1));
    MIH_TRY(nkey.alloc(G)); MIH_TRY(nkey2.alloc(G)); MIH_TRY(gid.alloc(G)); MIH_TRY(gid2.alloc(G)); MIH_TRY(rank.alloc(G));
    unsigned nb = (unsigned)((len + 255) / 256), gb = (unsigned)((G + 255) / 256);
    hipLaunchKernelGGL(k_grp_keys, dim3(nb), dim3(256), 0, s, y_dev, len, key.p, idx.p);
    size_t tmp_bytes = 0, need = 0;
    MIH_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, need, key.p, key2.p, idx.p, perm.p, (int)len, 0, 64, s));
    tmp_bytes = need;
    MIH_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, lab.p, lab2.p, perm.p, member.p, (int)len, 0, 32, s));
    if (need > tmp_bytes) tmp_bytes = need;
    MIH_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, need, nkey.p, nkey2.p, gid.p, gid2.p, (int)G, 0, 64, s));
    if (need > tmp_bytes) tmp_bytes = need;
    DevBuf<uint8_t> tmp;
    MIH_TRY(tmp.alloc(tmp_bytes));
    size_t tb = tmp_bytes;
    MIH_HIP(hipcub::DeviceRadixSort::SortPairsDescending(tmp.p, tb, key.p, key2.p, idx.p, perm.p, (int)len, 0, 64, s));
    hipLaunchKernelGGL(k_grp_labels, dim3(nb), dim3(256), 0, s, perm.p, group_dev, len, lab.p);
    tb = tmp_bytes;
    MIH_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, lab.p, lab2.p, perm.p, member.p, (int)len, 0, 32, s));
    hipLaunchKernelGGL(k_grp_bounds, dim3(nb), dim3(256), 0, s, lab2.p, len, G, seg.p);
    hipLaunchKernelGGL(k_grp_norms, dim3(gb), dim3(256), 0, s, y_dev, member.p, seg.p, G, k_dev, k_is_vector, nkey.p, gid.p);
    tb = tmp_bytes;
    MIH_HIP(hipcub::DeviceRadixSort::SortPairsDescending(tmp.p, tb, nkey.p, nkey2.p, gid.p, gid2.p, (int)G, 0, 64, s));
    hipLaunchKernelGGL(k_grp_rank, dim3(gb), dim3(256), 0, s, gid2.p, G, rank.p);
    hipLaunchKernelGGL(k_grp_apply, dim3(nb), dim3(256), 0, s, y_dev, member.p, lab2.p, seg.p, rank.p, len, J, k_dev, k_is_vector);
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
    DevBuf<double> dy; DevBuf<int64_t> dg, dk;
    MIH_TRY(dy.alloc(len)); MIH_TRY(dg.alloc(len)); MIH_TRY(dk.alloc(k_is_vector ? G : 1));
    MIH_HIP(hipMemcpy(dy.p, y, sizeof(double) * len, hipMemcpyHostToDevice));
    MIH_HIP(hipMemcpy(dg.p, group, sizeof(int64_t) * len, hipMemcpyHostToDevice));
    MIH_HIP(hipMemcpy(dk.p, k, sizeof(int64_t) * (k_is_vector ? G : 1), hipMemcpyHostToDevice));
    MIH_TRY(group_project_device(dy.p, dg.p, len, G, J, dk.p, k_is_vector, nullptr));
    MIH_HIP(hipMemcpy(y, dy.p, sizeof(double) * len, hipMemcpyDeviceToHost));
    return MIH_OK;
}
