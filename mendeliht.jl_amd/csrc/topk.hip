// topk.hip -- project_k!(x, k) on the device (src/utilities.jl:553-559):
// a = |k-th largest by magnitude|, then every |x_i| < a is zeroed; entries tied
// with a are KEPT.  Radix select over the IEEE-754 bit pattern of |x| (monotone for
// non-negative doubles, +Inf sorts last so the `zkeep` slots of vectorize!
// (utilities.jl:313-314) always survive): 8 passes of 8 bits with integer
// histograms (exact, order-independent => the selected support is reproducible),
// then one threshold pass that also compacts the survivors.
#include "common.h"
#include <algorithm>

namespace mih {

__device__ __forceinline__ uint64_t abs_key(double v)
{
    return (uint64_t)__double_as_longlong(v) & 0x7FFFFFFFFFFFFFFFull;
}

// state[0] = prefix (bits above `shift+8` already fixed), state[1] = remaining rank
__global__ void __launch_bounds__(256)
k_hist(const double *__restrict__ x, int64_t len, int shift, const uint64_t *__restrict__ state,
       uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t prefix = state[0];
    const bool first = (shift == 56);
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < len; i += stride) {
        uint64_t key = abs_key(x[i]);
        if (first || (key >> (shift + 8)) == prefix) atomicAdd(&h[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void k_pick(uint32_t *__restrict__ hist, uint64_t *__restrict__ state, int shift)
{
    if (threadIdx.x != 0) return;
    uint64_t kth = state[1], cum = 0;
    int bin = 255;
    for (; bin > 0; --bin) {
        if (cum + hist[bin] >= kth) break;
        cum += hist[bin];
    }
    state[0] = (shift == 56 ? 0ull : (state[0] << 8)) | (uint64_t)bin;
    state[1] = kth - cum;
    for (int b = 0; b < 256; ++b) hist[b] = 0;
    if (shift == 0) state[2] = state[0];   // full 64-bit threshold key
}

__global__ void __launch_bounds__(256)
k_threshold(double *__restrict__ x, int64_t len, const uint64_t *__restrict__ state,
            int64_t *__restrict__ sel_idx, double *__restrict__ sel_val, uint32_t *__restrict__ sel_cnt,
            uint32_t cap)
{
    const uint64_t thr = state[2];
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < len; i += stride) {
        double v = x[i];
        if (abs_key(v) < thr) { if (v != 0.0) x[i] = 0.0; }
        else if (v != 0.0) {
            uint32_t pos = atomicAdd(sel_cnt, 1u);
            if (pos < cap) { sel_idx[pos] = i; sel_val[pos] = v; }
        }
    }
}

int topk_work_init(TopkWork &w, int64_t max_keep)
{
    MIH_TRY(w.hist.alloc(256));
    MIH_TRY(w.state.alloc(4));
    MIH_TRY(w.sel_cnt.alloc(1));
    w.cap = max_keep + 1024;
    MIH_TRY(w.sel_idx.alloc((size_t)w.cap));
    MIH_TRY(w.sel_val.alloc((size_t)w.cap));
    return MIH_OK;
}

static int compact_device(double *x_dev, int64_t len, TopkWork &w, hipStream_t s,
                          std::vector<int64_t> &idx_out, std::vector<double> &val_out);

int topk_project_device(double *x_dev, int64_t len, int64_t k, TopkWork &w, hipStream_t s,
                        std::vector<int64_t> &idx_out, std::vector<double> &val_out)
{
    // utilities.jl:554 DomainError for k<0; partialsort(x, 0) / k>len is a BoundsError
    if (k <= 0 || k > len) { set_error("Attempted to project to sparsity level %lld (vector length %lld)", (long long)k, (long long)len); return MIH_BAD_ARG; }
    uint64_t st[4] = {0ull, (uint64_t)k, 0ull, 0ull};
    MIH_HIP(hipMemcpyAsync(w.state.p, st, sizeof(st), hipMemcpyHostToDevice, s));
    MIH_HIP(hipMemsetAsync(w.hist.p, 0, 256 * sizeof(uint32_t), s));
    int grid = (int)std::min<int64_t>((len + 255) / 256, 2048);
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), 0, s, x_dev, len, shift, w.state.p, w.hist.p);
        hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), 0, s, w.hist.p, w.state.p, shift);
    }
    return compact_device(x_dev, len, w, s, idx_out, val_out);
}

// threshold pass with the key already in state[2]: zero what is below, gather what survives
static int compact_device(double *x_dev, int64_t len, TopkWork &w, hipStream_t s,
                          std::vector<int64_t> &idx_out, std::vector<double> &val_out)
{
    int grid = (int)std::min<int64_t>((len + 255) / 256, 2048);
    for (int attempt = 0; attempt < 2; ++attempt) {
        MIH_HIP(hipMemsetAsync(w.sel_cnt.p, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_threshold, dim3(grid), dim3(256), 0, s, x_dev, len, w.state.p, w.sel_idx.p, w.sel_val.p,
                           w.sel_cnt.p, (uint32_t)w.cap);
        uint32_t cnt = 0;
        MIH_HIP(hipMemcpyAsync(&cnt, w.sel_cnt.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MIH_HIP(hipStreamSynchronize(s));
        if ((int64_t)cnt > w.cap) {            // many exact ties: grow and compact again
            w.cap = (int64_t)cnt + 1024;
            MIH_TRY(w.sel_idx.alloc((size_t)w.cap));
            MIH_TRY(w.sel_val.alloc((size_t)w.cap));
            continue;
        }
        std::vector<int64_t> ti(cnt); std::vector<double> tv(cnt);
        if (cnt) {
            MIH_HIP(hipMemcpy(ti.data(), w.sel_idx.p, sizeof(int64_t) * cnt, hipMemcpyDeviceToHost));
            MIH_HIP(hipMemcpy(tv.data(), w.sel_val.p, sizeof(double) * cnt, hipMemcpyDeviceToHost));
        }
        std::vector<uint32_t> ord(cnt);
        for (uint32_t i = 0; i < cnt; ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return ti[a] < ti[b]; });
        idx_out.resize(cnt); val_out.resize(cnt);
        for (uint32_t i = 0; i < cnt; ++i) { idx_out[i] = ti[ord[i]]; val_out[i] = tv[ord[i]]; }
        return MIH_OK;
    }
    set_error("top-k compaction failed");
    return MIH_HIP_ERROR;
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
