// fit_common.h -- small device helpers shared by the univariate (fit.hip) and multivariate
// (mv.hip) IHT drivers.  Kernels are `static` so each translation unit gets its own copy.
#pragma once
#include "common.h"
#include <vector>

namespace mih {

// deterministic block sum (fixed tree) of up to NV values per thread
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *out /* NV values, thread 0 */)
{
    __shared__ double red[NV][256];
    #pragma unroll
    for (int k = 0; k < NV; ++k) red[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            #pragma unroll
            for (int k = 0; k < NV; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        #pragma unroll
        for (int k = 0; k < NV; ++k) out[k] = red[k][0];
    }
}

// second stage: sum `nblocks` rows of NV partials in fixed order
static __global__ void k_final_sum(const double *__restrict__ partial, int nblocks, int nv, double *__restrict__ out)
{
    __shared__ double red[256];
    for (int k = 0; k < nv; ++k) {
        double a = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 256) a += partial[(int64_t)b * nv + k];
        red[threadIdx.x] = a;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
        if (threadIdx.x == 0) out[k] = red[0];
        __syncthreads();
    }
}

static __global__ void k_gather(const double *__restrict__ src, const int64_t *__restrict__ idx, int64_t nnz,
                         double *__restrict__ out)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < nnz) out[t] = src[idx[t]];
}
static __global__ void k_mask_to_wts(const uint8_t *__restrict__ m, int64_t n, int invert, double *__restrict__ w)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) w[i] = ((m[i] != 0) != (invert != 0)) ? 1.0 : 0.0;
}

static inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

struct Sparse {                       // a k-sparse p-vector, sorted by index
    std::vector<int64_t> idx;
    std::vector<double> val;
    void clear() { idx.clear(); val.clear(); }
};


}  // namespace mih
