// sweep_probe.hip -- what a small n-vector sweep costs inside a chain of kernels on MI355X (stand-alone probe, not part of the product).
// Round 5 asked why k_res_stats (44 MB of traffic, 500 k rows) takes 21 us and k_digits 15 us when their bytes are worth 5 us:
// one launch = NV input vectors of n doubles read, one written; variants in the mapping of rows to threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NV>
__global__ void __launch_bounds__(256) k_rowwise(const double *__restrict__ in, double *__restrict__ out, long n)
{
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    #pragma unroll
    for (int v = 0; v < NV; ++v) a += in[(long)v * n + i];
    out[i] = a;
}
// a strided walk by `blocks` workgroups, U rows in flight per thread, a sequential sum per thread and a block tree (k_r_stats's shape)
template <int NV, int U>
__global__ void __launch_bounds__(256) k_walk(const double *__restrict__ in, double *__restrict__ out, double *__restrict__ part, long n)
{
    __shared__ double sh[256];
    const long stride = 256l * gridDim.x;
    double s = 0.0;
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += U * stride) {
        double a[U];
        #pragma unroll
        for (int u = 0; u < U; ++u) {
            const long iu = i + u * stride;
            double t = 0.0;
            if (iu < n) {
                #pragma unroll
                for (int v = 0; v < NV; ++v) t += in[(long)v * n + iu];
            }
            a[u] = t;
        }
        #pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n) { out[i + u * stride] = a[u]; s += a[u]; }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ void k_empty(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

template <typename F> static float chain(F launch, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

// ... with the vectors COLD: a 1 GB buffer is rewritten before every timed launch (in the product the 125 GB X'r pass sits between
// two step chains: nothing of the fit's vectors survives in L2, the 256 MB MALL or the TLBs)
static char *g_evict; static size_t g_evict_bytes = 1ull << 30;
template <typename F> static float cold(F launch, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0.f;
    for (int i = 0; i < reps; ++i) {
        CK(hipMemsetAsync(g_evict, i, g_evict_bytes, 0));
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
    }
    return tot * 1e3f / reps;
}

int main()
{
    const long n = 500000;
    double *in, *out, *part; CK(hipMalloc((void **)&in, sizeof(double) * n * 8)); CK(hipMalloc((void **)&out, sizeof(double) * n)); CK(hipMalloc((void **)&part, 8 * 4096));
    CK(hipMemset(in, 0, sizeof(double) * n * 8));
    // a 256 MB buffer written between the launches of a pair evicts the vectors from L2 / MALL as the X'r pass does in the product
    const int nb = (int)((n + 255) / 256);
    printf("empty kernel                       : %6.2f us per launch\n", chain([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, (int *)nullptr); }, 200));
    printf("row-wise 1 in, %4d blocks         : %6.2f us\n", nb, chain([&] { hipLaunchKernelGGL(k_rowwise<1>, dim3(nb), dim3(256), 0, 0, in, out, n); }, 200));
    printf("row-wise 5 in, %4d blocks         : %6.2f us\n", nb, chain([&] { hipLaunchKernelGGL(k_rowwise<5>, dim3(nb), dim3(256), 0, 0, in, out, n); }, 200));
    for (int blocks : {64, 128, 256, 512, 1024}) {
        printf("walk 5 in, %4d blocks, 1 in flight : %6.2f us\n", blocks, chain([&] { hipLaunchKernelGGL((k_walk<5, 1>), dim3(blocks), dim3(256), 0, 0, in, out, part, n); }, 200));
        printf("walk 5 in, %4d blocks, 4 in flight : %6.2f us\n", blocks, chain([&] { hipLaunchKernelGGL((k_walk<5, 4>), dim3(blocks), dim3(256), 0, 0, in, out, part, n); }, 200));
        printf("walk 5 in, %4d blocks, 16 in flight: %6.2f us\n", blocks, chain([&] { hipLaunchKernelGGL((k_walk<5, 16>), dim3(blocks), dim3(256), 0, 0, in, out, part, n); }, 200));
    }
    printf("walk 1 in,   64 blocks, 1 in flight : %6.2f us\n", chain([&] { hipLaunchKernelGGL((k_walk<1, 1>), dim3(64), dim3(256), 0, 0, in, out, part, n); }, 200));
    printf("walk 1 in,   64 blocks, 16 in flight: %6.2f us\n", chain([&] { hipLaunchKernelGGL((k_walk<1, 16>), dim3(64), dim3(256), 0, 0, in, out, part, n); }, 200));
    printf("walk 1 in,   64 blocks, 32 in flight: %6.2f us\n", chain([&] { hipLaunchKernelGGL((k_walk<1, 32>), dim3(64), dim3(256), 0, 0, in, out, part, n); }, 200));
    {   // the same chain of small kernels as a hipGraph: does a captured chain have cheaper boundaries than stream launches the host queues ahead?
        hipStream_t st; CK(hipStreamCreate(&st));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 13; ++i) hipLaunchKernelGGL(k_rowwise<1>, dim3(nb), dim3(256), 0, st, in, out, n);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("13 row-wise kernels as a hipGraph   : %6.2f us per kernel (graph launches back to back)\n", ms * 1e3f / (50 * 13));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 50 * 13; ++i) hipLaunchKernelGGL(k_rowwise<1>, dim3(nb), dim3(256), 0, st, in, out, n);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("the same 650 kernels, stream launches: %6.2f us per kernel\n", ms * 1e3f / (50 * 13));
    }
    CK(hipMalloc((void **)&g_evict, g_evict_bytes));
    printf("-- cold (1 GB rewritten before each launch; the event pair's own cost is in every line) --\n");
    printf("empty kernel                        : %6.2f us\n", cold([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, (int *)nullptr); }, 30));
    printf("row-wise 1 in, %4d blocks          : %6.2f us\n", nb, cold([&] { hipLaunchKernelGGL(k_rowwise<1>, dim3(nb), dim3(256), 0, 0, in, out, n); }, 30));
    printf("row-wise 5 in, %4d blocks          : %6.2f us\n", nb, cold([&] { hipLaunchKernelGGL(k_rowwise<5>, dim3(nb), dim3(256), 0, 0, in, out, n); }, 30));
    for (int blocks : {64, 128, 256, 512}) {
        printf("walk 5 in, %4d blocks, 1 in flight : %6.2f us\n", blocks, cold([&] { hipLaunchKernelGGL((k_walk<5, 1>), dim3(blocks), dim3(256), 0, 0, in, out, part, n); }, 30));
        printf("walk 5 in, %4d blocks, 16 in flight: %6.2f us\n", blocks, cold([&] { hipLaunchKernelGGL((k_walk<5, 16>), dim3(blocks), dim3(256), 0, 0, in, out, part, n); }, 30));
    }
    printf("row-wise 5 in twice in a row (second warm): %6.2f us for both\n", cold([&] { hipLaunchKernelGGL(k_rowwise<5>, dim3(nb), dim3(256), 0, 0, in, out, n); hipLaunchKernelGGL(k_rowwise<5>, dim3(nb), dim3(256), 0, 0, in, out, n); }, 30));
    return 0;
}
