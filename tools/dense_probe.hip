// dense_probe.hip -- stand-alone probe (not part of the product): dense column-dot kernels with different numbers of
// loads in flight, waves per column, and the residual chunk shared through LDS (hipcc --offload-arch=gfx950 -O3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double wave_sum(double v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64); return v; }

template <int U, int WPC>   // U loads in flight per lane, WPC waves per column
__global__ void __launch_bounds__(256) k_dense(const double *__restrict__ D, int64_t n, int64_t p, const double *__restrict__ r, double *__restrict__ out)
{
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j = (blockIdx.x * 4ll + wave) / WPC;
    const int sub = (int)((blockIdx.x * 4ll + wave) % WPC);
    if (j >= p) return;
    const f64x2 *cx = reinterpret_cast<const f64x2 *>(D + j * n);
    const f64x2 *rx = reinterpret_cast<const f64x2 *>(r);
    const int64_t n2 = n >> 1;
    const int64_t seg = ((n2 + WPC - 1) / WPC + 63) / 64 * 64;
    const int64_t lo = sub * seg, hi = (lo + seg < n2) ? lo + seg : n2;
    double a[2 * U];
    for (int t = 0; t < 2 * U; ++t) a[t] = 0.0;
    int64_t i = lo + lane;
    for (; i + 64 * (U - 1) < hi; i += 64 * U) {
        f64x2 x[U], v[U];
        #pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(cx + i + 64 * u);
        #pragma unroll
        for (int u = 0; u < U; ++u) v[u] = rx[i + 64 * u];
        #pragma unroll
        for (int u = 0; u < U; ++u) { a[2 * u] = fma(x[u].x, v[u].x, a[2 * u]); a[2 * u + 1] = fma(x[u].y, v[u].y, a[2 * u + 1]); }
    }
    for (; i < hi; i += 64) { f64x2 x0 = __builtin_nontemporal_load(cx + i), v0 = rx[i]; a[0] = fma(x0.x, v0.x, a[0]); a[1] = fma(x0.y, v0.y, a[1]); }
    double s = 0.0;
    for (int t = 0; t < 2 * U; ++t) s += a[t];
    s = wave_sum(s);
    if (WPC == 1) { if (lane == 0) out[j] = s; }
    else {
        if (lane == 0) part[wave] = s;
        __syncthreads();
        if (lane == 0 && sub == 0) { double t = 0.0; for (int w = 0; w < WPC; ++w) t += part[wave + w]; out[j] = t; }
    }
}

// residual chunk shared through LDS by the WAVES columns of a block (one barrier per 4 KB step, double buffered)
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_dense_lds(const double *__restrict__ D, int64_t n, int64_t p, const double *__restrict__ r, double *__restrict__ out)
{
    __shared__ f64x2 rt[2][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t j = blockIdx.x * (int64_t)WAVES + wave;
    if (j >= p) j = p - 1;
    const f64x2 *cx = reinterpret_cast<const f64x2 *>(D + j * n);
    const f64x2 *rx = reinterpret_cast<const f64x2 *>(r);
    const int64_t n2 = n >> 1;
    const int64_t steps = (n2 + 255) / 256;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // stage step 0
    for (int t = threadIdx.x; t < 256; t += WAVES * 64) rt[0][t] = (t < n2) ? rx[t] : f64x2{0.0, 0.0};
    __syncthreads();
    f64x2 x[4];
    #pragma unroll
    for (int u = 0; u < 4; ++u) { int64_t i = lane + 64 * u; x[u] = (i < n2) ? __builtin_nontemporal_load(cx + i) : f64x2{0.0, 0.0}; }
    for (int64_t st = 0; st < steps; ++st) {
        const int buf = (int)(st & 1);
        const int64_t base = (st + 1) * 256;
        f64x2 rn = {0.0, 0.0};
        if (threadIdx.x < 256 && base + threadIdx.x < n2) rn = rx[base + threadIdx.x];
        f64x2 xn[4];
        #pragma unroll
        for (int u = 0; u < 4; ++u) { int64_t i = base + lane + 64 * u; xn[u] = (i < n2) ? __builtin_nontemporal_load(cx + i) : f64x2{0.0, 0.0}; }
        #pragma unroll
        for (int u = 0; u < 4; ++u) { f64x2 v = rt[buf][lane + 64 * u]; a[2 * u] = fma(x[u].x, v.x, a[2 * u]); a[2 * u + 1] = fma(x[u].y, v.y, a[2 * u + 1]); }
        if (threadIdx.x < 256) rt[buf ^ 1][threadIdx.x] = rn;
        __syncthreads();
        #pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = xn[u];
    }
    double s = wave_sum(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
    if (lane == 0 && blockIdx.x * (int64_t)WAVES + wave < p) out[j] = s;
}
template <int WAVES> static void run_lds(const double *D, int64_t n, int64_t p, const double *r, double *out, const char *name)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((p + WAVES - 1) / WAVES);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_dense_lds<WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, D, n, p, r, out);
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k_dense_lds<WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, D, n, p, r, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    double B = 8.0 * n * p + 8.0 * (n + p);
    printf("%-22s %.3f ms  %.0f GB/s (%.1f%%)\n", name, ms, B / ms / 1e6, B / ms / 1e6 / 80);
}

template <int U, int WPC> static void run(const double *D, int64_t n, int64_t p, const double *r, double *out, const char *name)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((p * WPC + 3) / 4);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_dense<U, WPC>), dim3(grid), dim3(256), 0, 0, D, n, p, r, out);
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k_dense<U, WPC>), dim3(grid), dim3(256), 0, 0, D, n, p, r, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    double B = 8.0 * n * p + 8.0 * (n + p);
    printf("%-22s %.3f ms  %.0f GB/s (%.1f%%)\n", name, ms, B / ms / 1e6, B / ms / 1e6 / 80);
}

int main()
{
    const int64_t n = 50000, p = 100000;
    double *D, *r, *out;
    CK(hipMalloc((void **)&D, sizeof(double) * n * p)); CK(hipMalloc((void **)&r, sizeof(double) * n)); CK(hipMalloc((void **)&out, sizeof(double) * p));
    CK(hipMemset(D, 0, sizeof(double) * n * p)); CK(hipMemset(r, 0, sizeof(double) * n));
    run<4, 1>(D, n, p, r, out, "U=4 1 wave/col");
    run_lds<4>(D, n, p, r, out, "lds 4 waves");
    run_lds<8>(D, n, p, r, out, "lds 8 waves");
    run_lds<16>(D, n, p, r, out, "lds 16 waves");
    run<4, 1>(D, n, p, r, out, "U=4 1 wave/col");
    return 0;
}
