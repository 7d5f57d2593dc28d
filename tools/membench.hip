// membench.hip -- which access shape streams a column-major 2-bit matrix fastest on MI355X?
// Stand-alone probe (not part of the product): reads only, XOR-reduces what it loads.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o gpurun_out/membench && ./membench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int LW> struct V;
template <> struct V<1> { using t = uint32_t; };
template <> struct V<2> { using t = uint2; };
template <> struct V<4> { using t = uint4; };
__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// Shape A: a wave owns C columns and walks a row slice; DEPTH superchunks in flight per column.
template <int WAVES, int C, int LW, int DEPTH>
__global__ void __launch_bounds__(WAVES * 64)
k_colwave(const uint32_t *__restrict__ X, int64_t stride_dw, int64_t p, int64_t nsc, int splits, uint32_t *out)
{
    using T = typename V<LW>::t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int split = blockIdx.x % splits;
    const int64_t cg = blockIdx.x / splits;
    const int64_t sps = (nsc + splits - 1) / splits;
    const int64_t c0 = split * sps, c1 = (c0 + sps < nsc) ? c0 + sps : nsc;
    const int64_t j0 = cg * (WAVES * C) + wave * C;
    const T *col[C];
    #pragma unroll
    for (int c = 0; c < C; ++c) { int64_t j = j0 + c < p ? j0 + c : p - 1; col[c] = reinterpret_cast<const T *>(X + j * stride_dw) + lane; }
    uint32_t acc = 0;
    T buf[DEPTH][C];
    #pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        #pragma unroll
        for (int c = 0; c < C; ++c) { int64_t s = c0 + d < c1 ? c0 + d : c1 - 1; buf[d][c] = col[c][s * 64]; }
    for (int64_t s = c0; s < c1; s += DEPTH) {
        #pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            #pragma unroll
            for (int c = 0; c < C; ++c) {
                acc ^= fold(buf[d][c]);
                int64_t sn = s + d + DEPTH; if (sn >= c1) sn = c1 - 1;
                buf[d][c] = col[c][sn * 64];
            }
        }
    }
    #pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        #pragma unroll
        for (int c = 0; c < C; ++c) acc ^= fold(buf[d][c]);
    if (acc == 0x12345678u) out[0] = acc;
}

// Shape B: the WAVES waves of a workgroup read adjacent segments of the SAME column (the workgroup
// owns C columns; wave w takes superchunks w, w+WAVES, ...): WAVES*256*LW contiguous bytes per column per step.
template <int WAVES, int C, int LW, int DEPTH>
__global__ void __launch_bounds__(WAVES * 64)
k_colblock(const uint32_t *__restrict__ X, int64_t stride_dw, int64_t p, int64_t nsc, int splits, uint32_t *out)
{
    using T = typename V<LW>::t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int split = blockIdx.x % splits;
    const int64_t cg = blockIdx.x / splits;
    const int64_t sps = (nsc + splits - 1) / splits;
    const int64_t c0 = split * sps, c1 = (c0 + sps < nsc) ? c0 + sps : nsc;
    const int64_t j0 = cg * C;
    const T *col[C];
    #pragma unroll
    for (int c = 0; c < C; ++c) { int64_t j = j0 + c < p ? j0 + c : p - 1; col[c] = reinterpret_cast<const T *>(X + j * stride_dw) + lane; }
    uint32_t acc = 0;
    T buf[DEPTH][C];
    #pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        #pragma unroll
        for (int c = 0; c < C; ++c) { int64_t s = c0 + wave + d * WAVES; if (s >= c1) s = c1 - 1; buf[d][c] = col[c][s * 64]; }
    for (int64_t s = c0 + wave; s < c1; s += DEPTH * WAVES) {
        #pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            #pragma unroll
            for (int c = 0; c < C; ++c) {
                acc ^= fold(buf[d][c]);
                int64_t sn = s + (d + DEPTH) * WAVES; if (sn >= c1) sn = c1 - 1;
                buf[d][c] = col[c][sn * 64];
            }
        }
    }
    #pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        #pragma unroll
        for (int c = 0; c < C; ++c) acc ^= fold(buf[d][c]);
    if (acc == 0x12345678u) out[0] = acc;
}

// Shape L: plain linear streaming read of the whole buffer (the ceiling).
__global__ void __launch_bounds__(256) k_linear(const uint4 *__restrict__ X, int64_t n16, uint32_t *out)
{
    int64_t i = blockIdx.x * 256ll + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = X[i], b = X[i + stride], c = X[i + 2 * stride], d = X[i + 3 * stride];
        acc ^= fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
    }
    for (; i < n16; i += stride) acc ^= fold(X[i]);
    if (acc == 0x12345678u) out[0] = acc;
}

static uint32_t *X, *out;
static int64_t stride_dw, p;
static double bytes;

template <typename F> static void timeit(const char *name, F launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 3; ++i) launch(); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    printf("%-44s %8.3f ms %8.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout);
}

template <int WAVES, int C, int LW, int DEPTH> static void runA(int splits)
{
    int64_t nsc = stride_dw / (64 * LW);
    int64_t groups = (p + WAVES * C - 1) / (WAVES * C);
    char name[128]; snprintf(name, sizeof name, "A colwave  W%d C%-2d LW%d depth%d splits%d", WAVES, C, LW, DEPTH, splits);
    timeit(name, [&] { hipLaunchKernelGGL((k_colwave<WAVES, C, LW, DEPTH>), dim3((unsigned)(groups * splits)), dim3(WAVES * 64), 0, 0, X, stride_dw, p, nsc, splits, out); });
}
template <int WAVES, int C, int LW, int DEPTH> static void runB(int splits)
{
    int64_t nsc = stride_dw / (64 * LW);
    int64_t groups = (p + C - 1) / C;
    char name[128]; snprintf(name, sizeof name, "B colblock W%d C%-2d LW%d depth%d splits%d", WAVES, C, LW, DEPTH, splits);
    timeit(name, [&] { hipLaunchKernelGGL((k_colblock<WAVES, C, LW, DEPTH>), dim3((unsigned)(groups * splits)), dim3(WAVES * 64), 0, 0, X, stride_dw, p, nsc, splits, out); });
}

int main(int argc, char **argv)
{
    int64_t n = 500000; p = argc > 1 ? atoll(argv[1]) : 262144;
    stride_dw = ((n + 15) / 16 + 255) / 256 * 256;
    bytes = (double)p * stride_dw * 4;
    CK(hipMalloc((void **)&X, (size_t)bytes)); CK(hipMalloc((void **)&out, 64));
    CK(hipMemset(X, 0x5a, (size_t)bytes));
    printf("matrix: p=%lld columns x %lld B = %.1f GB\n", (long long)p, (long long)stride_dw * 4, bytes / 1e9);
    timeit("L linear uint4 grid-stride (2048 blocks)", [&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, (const uint4 *)X, (int64_t)(bytes / 16), out); });
    timeit("L linear uint4 grid-stride (8192 blocks)", [&] { hipLaunchKernelGGL(k_linear, dim3(8192), dim3(256), 0, 0, (const uint4 *)X, (int64_t)(bytes / 16), out); });
    runA<4, 8, 1, 1>(8); runA<4, 8, 1, 2>(8); runA<4, 8, 1, 4>(8); runA<4, 8, 1, 2>(1);
    runA<4, 4, 1, 4>(8); runA<4, 2, 1, 8>(8); runA<4, 1, 1, 8>(8); runA<4, 16, 1, 2>(8);
    runA<4, 4, 2, 2>(8); runA<4, 4, 2, 4>(8); runA<4, 8, 2, 2>(8);
    runA<4, 4, 4, 1>(8); runA<4, 4, 4, 2>(8); runA<4, 2, 4, 4>(8); runA<4, 1, 4, 8>(8); runA<4, 8, 4, 1>(8); runA<4, 4, 4, 2>(1);
    runA<8, 4, 4, 2>(8); runA<8, 2, 4, 2>(8); runA<16, 2, 4, 2>(8);
    runB<4, 8, 1, 2>(8); runB<4, 8, 1, 4>(8); runB<4, 4, 4, 2>(8); runB<4, 8, 4, 1>(8); runB<4, 8, 4, 2>(8); runB<8, 8, 1, 2>(8); runB<8, 8, 4, 1>(8);
    runB<4, 8, 4, 2>(1); runB<4, 16, 1, 2>(8); runB<16, 4, 1, 2>(8);
    return 0;
}
