// gridsync_probe.hip -- what does a grid-wide barrier cost on MI355X (8 XCDs, 256 CUs)?  Decides whether the kernels of a resident IHT
// step can become phases of ONE persistent kernel.  hipcc --offload-arch=gfx950 -O3 tools/gridsync_probe.hip -o /tmp/gridsync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_cg(int iters, double *sink)
{
    cg::grid_group g = cg::this_grid();
    double a = threadIdx.x;
    for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + 1.0; g.sync(); }
    if (a == 12345.678) *sink = a;
}
// one counter, monotone generation: block b arrives (atomic add), the last one bumps gen; everyone spins on gen
__device__ __forceinline__ void bar_flat(unsigned *cnt, unsigned *gen, unsigned nblocks, unsigned &my_gen)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        ++my_gen;
        __threadfence();
        if (atomicAdd(cnt, 1u) == nblocks - 1) { *cnt = 0; __threadfence(); __hip_atomic_store(gen, my_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        else while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < my_gen) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}
__global__ void k_flat(int iters, unsigned *cnt, unsigned *gen, double *sink)
{
    unsigned my_gen = 0;
    double a = threadIdx.x;
    for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + 1.0; bar_flat(cnt, gen, gridDim.x, my_gen); }
    if (a == 12345.678) *sink = a;
}
// two levels: the blocks with the same blockIdx % 8 (one XCD, if the dispatcher deals blocks out round-robin) meet at their own counter
// (own cache line), the eight last arrivals meet at the global one
__global__ void k_two(int iters, unsigned *cnt8 /* 8 x 32 */, unsigned *cnt, unsigned *gen, double *sink)
{
    unsigned my_gen = 0;
    const unsigned x = blockIdx.x & 7, per = (gridDim.x + 7 - x) / 8;
    double a = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        a = a * 1.0000001 + 1.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            ++my_gen;
            __threadfence();
            bool release = false;
            if (atomicAdd(&cnt8[32 * x], 1u) == per - 1) {
                cnt8[32 * x] = 0;
                if (atomicAdd(cnt, 1u) == 7) { *cnt = 0; release = true; }
            }
            if (release) { __threadfence(); __hip_atomic_store(gen, my_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
            else while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < my_gen) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
    if (a == 12345.678) *sink = a;
}
__global__ void k_empty(double *sink) { if (threadIdx.x == 9999) *sink = 1.0; }

int main()
{
    unsigned *cnt, *gen, *cnt8; double *sink;
    CK(hipMalloc(&cnt, 256)); CK(hipMalloc(&gen, 256)); CK(hipMalloc(&cnt8, 8 * 32 * 4)); CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200;
    for (int threads : {256, 512, 1024}) for (int blocks : {256, 512, 1024}) {
        if ((long)blocks * threads > 256l * 2048) continue;
        float ms;
        CK(hipMemset(cnt, 0, 256)); CK(hipMemset(gen, 0, 256)); CK(hipMemset(cnt8, 0, 8 * 32 * 4));
        // cooperative groups
        int it = iters; void *args[] = {&it, &sink};
        hipError_t e = hipLaunchCooperativeKernel((void *)k_cg, dim3(blocks), dim3(threads), args, 0, 0);
        if (e == hipSuccess) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchCooperativeKernel((void *)k_cg, dim3(blocks), dim3(threads), args, 0, 0));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("blocks %4d x %4d threads: cooperative_groups grid.sync  %7.2f us per barrier\n", blocks, threads, 1e3 * ms / iters);
        } else { printf("blocks %d x %d: cooperative launch refused (%s)\n", blocks, threads, hipGetErrorString(e)); (void)hipGetLastError(); }
        void *a2[] = {&it, &cnt, &gen, &sink};
        CK(hipLaunchCooperativeKernel((void *)k_flat, dim3(blocks), dim3(threads), a2, 0, 0)); CK(hipDeviceSynchronize());
        CK(hipMemset(gen, 0, 256));
        CK(hipEventRecord(e0, 0));
        CK(hipLaunchCooperativeKernel((void *)k_flat, dim3(blocks), dim3(threads), a2, 0, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("blocks %4d x %4d threads: one counter                   %7.2f us per barrier\n", blocks, threads, 1e3 * ms / iters);
        CK(hipMemset(gen, 0, 256));
        void *a3[] = {&it, &cnt8, &cnt, &gen, &sink};
        CK(hipLaunchCooperativeKernel((void *)k_two, dim3(blocks), dim3(threads), a3, 0, 0)); CK(hipDeviceSynchronize());
        CK(hipMemset(gen, 0, 256));
        CK(hipEventRecord(e0, 0));
        CK(hipLaunchCooperativeKernel((void *)k_two, dim3(blocks), dim3(threads), a3, 0, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("blocks %4d x %4d threads: per-XCD counters + one        %7.2f us per barrier\n", blocks, threads, 1e3 * ms / iters);
    }
    // back-to-back empty kernels: the cost of a kernel boundary
    for (int blocks : {1, 256, 2048}) {
        float ms;
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel, %4d blocks, back to back: %7.2f us per launch\n", blocks, 1e3 * ms / 200);
    }
    return 0;
}
