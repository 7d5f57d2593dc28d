// mfma_rate.hip -- what rate of v_mfma_f32_32x32x64_f8f6f4 (A = FP4 dosage-like, B = FP4 or FP6 digit-like) does the
// whole chip sustain, on zero and on random operands, with the operands in registers (no memory traffic)?  The
// in-kernel clock is s_memtime / s_memrealtime (100 MHz).  Stand-alone probe: the fused X'R pass issues 9.77e8 of these
// per 12 residuals at n = 500k, p = 1M, so this rate bounds that pass from below whatever the memory system does.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o build/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool FP6>
__global__ void __launch_bounds__(512)
k_rate(const uint32_t *__restrict__ opnd, int iters, float *__restrict__ sink, unsigned long long *__restrict__ stamps)
{
    const int lane = threadIdx.x & 63;
    i32x8 a[4], b[2];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) a[q][i] = i < 4 ? (int)opnd[(q * 8 + i) * 64 + lane] : 0;
    for (int q = 0; q < 2; ++q) for (int i = 0; i < 8; ++i) b[q][i] = i < (FP6 ? 6 : 4) ? (int)opnd[((4 + q) * 8 + i) * 64 + lane] : 0;
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int g = 0; g < 16; ++g) acc[k][g] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int k = 0; k < NACC; ++k) {
            if (FP6) acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[k & 3], b[(k >> 2) & 1], acc[k], 4, 2, 0, 0, 0, 0);
            else     acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[k & 3], b[(k >> 2) & 1], acc[k], 4, 4, 0, 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int g = 0; g < 16; ++g) s += acc[k][g];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) { const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = r1 - r0; }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the 16x16x128 form of the same instruction family: half the MACs per instruction (16 SNPs x 16 digit columns x 128 rows)
template <int NACC>
__global__ void __launch_bounds__(512)
k_rate16(const uint32_t *__restrict__ opnd, int iters, float *__restrict__ sink, unsigned long long *__restrict__ stamps)
{
    const int lane = threadIdx.x & 63;
    i32x8 a[4], b[2];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) a[q][i] = i < 4 ? (int)opnd[(q * 8 + i) * 64 + lane] : 0;
    for (int q = 0; q < 2; ++q) for (int i = 0; i < 8; ++i) b[q][i] = i < 6 ? (int)opnd[((4 + q) * 8 + i) * 64 + lane] : 0;
    f32x4 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int g = 0; g < 4; ++g) acc[k][g] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int k = 0; k < NACC; ++k)
            acc[k] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[k & 3], b[(k >> 2) & 1], acc[k], 4, 2, 0, 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int g = 0; g < 4; ++g) s += acc[k][g];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) { const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = r1 - r0; }
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    // (round 6) mfma_rate ITERS DATA SHAPE SECONDS: only that operand data and instruction shape, launched back to back for SECONDS
    // (tools/energy_table.py samples the package power beside it)
    const int only_data = argc > 2 ? atoi(argv[2]) : -1, only_shape = argc > 3 ? atoi(argv[3]) : -1;
    const double seconds = argc > 4 ? atof(argv[4]) : 0.0;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    uint32_t *opnd; float *sink; unsigned long long *stamps;
    CK(hipMalloc(&opnd, 6 * 8 * 64 * 4)); CK(hipMalloc(&sink, (size_t)cus * 512 * 4 * 4)); CK(hipMalloc(&stamps, (size_t)cus * 8 * 16 * 4));
    static const int E2M3U[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32};
    for (int data = 0; data < 3; ++data) {          // 0 zeros, 1 dosage-like A x random digits B, 2 dense random A x random digits
        if (only_data >= 0 && data != only_data) continue;
        std::vector<uint32_t> h(6 * 8 * 64, 0u);
        if (data) {
            for (int q = 0; q < 4; ++q) for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) {
                uint32_t w = 0;
                for (int e = 0; e < 8; ++e) {
                    uint32_t u = rnd() % 100, g = data == 2 ? rnd() % 3 : (u < 58 ? 0 : u < 91 ? 1 : 2);   // genotype frequencies of maf ~ U(0, 0.5)
                    w |= g << (4 * e);
                }
                h[(q * 8 + i) * 64 + l] = w;
            }
            for (int q = 0; q < 2; ++q) for (int l = 0; l < 64; ++l) {
                unsigned long long bits[3] = {0, 0, 0};           // 32 FP6 digit codes of the base-49 digit set
                for (int e = 0; e < 32; ++e) {
                    int r = (int)(rnd() % 49), d = r <= 16 ? r : r >= 33 ? r - 49 : (r & 1) ? r - 49 : r, u = abs(d);
                    int ex, m; if (u < 8) { ex = 0; m = u; } else if (u < 16) { ex = 1; m = u - 8; } else if (u <= 30) { ex = 2; m = u / 2 - 8; } else { ex = 3; m = u / 4 - 8; }
                    unsigned long long code = ((d < 0) << 5) | (ex << 3) | m;
                    int bit = 6 * e, wi = bit >> 6, sh = bit & 63;
                    bits[wi] |= code << sh; if (sh > 58) bits[wi + 1] |= code >> (64 - sh);
                }
                for (int i = 0; i < 6; ++i) h[((4 + q) * 8 + i) * 64 + l] = (uint32_t)(bits[i >> 1] >> (32 * (i & 1)));
            }
        }
        (void)E2M3U;
        CK(hipMemcpy(opnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int shape = 0; shape < 6; ++shape) {   // (waves per SIMD, FP6?, 16x16x128?)
            if (only_shape >= 0 && shape != only_shape) continue;
            const int wps = shape & 1 ? 2 : 1; const bool fp6 = shape < 2 || shape >= 4; const bool s16 = shape >= 4;
            const int threads = 256 * wps, blocks = cus;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e30f; double clk = 0;
            hipEvent_t w0, w1; CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1)); CK(hipEventRecord(w0));
            for (int rep = 0; rep < 6 || seconds > 0.0; ++rep) {      // ~1 s of back-to-back launches: the clock settles under load
                if (seconds > 0.0 && rep >= 6) { CK(hipEventRecord(w1)); CK(hipEventSynchronize(w1)); float el; CK(hipEventElapsedTime(&el, w0, w1)); if (el > 1e3 * seconds) break; }
                CK(hipEventRecord(e0));
                if (s16) hipLaunchKernelGGL((k_rate16<16>), dim3(blocks), dim3(threads), 0, 0, opnd, 2 * iters, sink, stamps);   // same MACs per launch
                else if (fp6) hipLaunchKernelGGL((k_rate<8, true>), dim3(blocks), dim3(threads), 0, 0, opnd, iters, sink, stamps);
                else     hipLaunchKernelGGL((k_rate<8, false>), dim3(blocks), dim3(threads), 0, 0, opnd, iters, sink, stamps);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 3 && ms < best) best = ms;
                std::vector<unsigned long long> st((size_t)blocks * 4 * wps * 2);
                CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
                double c = 0, r = 0; for (size_t w = 0; w < st.size() / 2; ++w) { c += st[2 * w]; r += st[2 * w + 1]; }
                clk = c / r * 100.0;                 // MHz
            }
            const double nm = (double)blocks * 4 * wps * 8.0 * iters;      // in units of one 32x32x64 (= two 16x16x128, launched 4x as many)
            printf("data %d (%s) %s %d wave/SIMD: %8.2f ms  %.3e MFMA/s  in-kernel clock %.0f MHz  -> 9.77e8 MFMAs (12 residuals, n=500k, p=1M) = %.1f ms; cycles/MFMA/SIMD %.1f\n",
                   data, data == 0 ? "zeros" : data == 1 ? "dosage x digits" : "dense x digits", s16 ? "FP4xFP6 16x16x128" : fp6 ? "FP4xFP6" : "FP4xFP4", wps,
                   best, nm / (best * 1e-3), clk, 9.77e8 / (nm / (best * 1e-3)) * 1e3, clk * 1e6 * best * 1e-3 / (8.0 * iters * wps));
        }
    }
    return 0;
}
