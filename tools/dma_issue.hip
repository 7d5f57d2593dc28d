// dma_issue.hip -- what does a wave pay to ISSUE its share of the loads of a fused X'R step beside 32 MFMAs?
// One wave per SIMD (256-thread workgroups, one per CU) runs `iters` steps of 32 v_mfma_f32_32x32x64_f8f6f4 (FP4 x FP6,
// random operands in registers) plus K loads of 1 KB per wave, in one of these forms:
//   mode 0  no loads
//   mode 1  K global_load_lds_dwordx4 (LDS-DMA) in a burst at the top of the step
//   mode 2  K LDS-DMA spread: one after every 32/K MFMAs
//   mode 3  K global_load_dwordx4 to VGPRs in a burst (consumed a step later)
//   mode 4  K global_load_dwordx4 spread
// src: 0 = a 4 MB buffer per XCD-ish region (L2 hits), 1 = a 16 GB stream (HBM).  Cycles per step from s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 tools/dma_issue.hip -o build/dma_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(uint32_t lds_dst, uint32_t voff, const void *sbase)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

template <int MODE, int K>
__global__ void __launch_bounds__(256, 1)
k_issue(const uint32_t *__restrict__ opnd, const char *__restrict__ src, size_t span, int iters, float *__restrict__ sink,
        unsigned long long *__restrict__ stamps)
{
    __shared__ uint4 lds[4 * 8192 / 16];          // 8 x 1 KB ring per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    i32x8 a[4], b[2];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) a[q][i] = i < 4 ? (int)opnd[(q * 8 + i) * 64 + lane] : 0;
    for (int q = 0; q < 2; ++q) for (int i = 0; i < 8; ++i) b[q][i] = i < 6 ? (int)opnd[((4 + q) * 8 + i) * 64 + lane] : 0;
    f32x16 acc[8];
    for (int k = 0; k < 8; ++k) for (int g = 0; g < 16; ++g) acc[k][g] = 0.f;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds + wave * 8192;
    const uint32_t voff = lane * 16;
    size_t pos = ((size_t)blockIdx.x * 4 + wave) * (size_t)(K > 0 ? K : 1) * 1024 * 64 % span;     // every wave its own stream
    u32x4 r[K > 0 ? K : 1];
    for (int k = 0; k < (K > 0 ? K : 1); ++k) r[k] = u32x4{0, 0, 0, 0};
    unsigned x = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3 || MODE == 4) {
            #pragma unroll
            for (int k = 0; k < K; ++k) x ^= r[k][0] ^ r[k][1] ^ r[k][2] ^ r[k][3];        // consume last step's loads
        }
        if (MODE == 1) {
            #pragma unroll
            for (int k = 0; k < K; ++k) glds16(lds0 + (k & 7) * 1024, voff, src + pos + (size_t)k * 1024);
        }
        if (MODE == 3) {
            #pragma unroll
            for (int k = 0; k < K; ++k) r[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + pos + (size_t)k * 1024) + lane);
        }
        #pragma unroll
        for (int m = 0; m < 32; ++m) {
            acc[m & 7] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m & 3], b[(m >> 2) & 1], acc[m & 7], 4, 2, 0, 0, 0, 0);
            if (K > 0 && (m % (32 / (K > 0 ? K : 1))) == 0 && m / (32 / (K > 0 ? K : 1)) < K) {
                const int k = m / (32 / (K > 0 ? K : 1));
                if (MODE == 2) glds16(lds0 + (k & 7) * 1024, voff, src + pos + (size_t)k * 1024);
                if (MODE == 4) r[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + pos + (size_t)k * 1024) + lane);
            }
        }
        pos += (size_t)K * 1024;
        if (pos + (size_t)K * 1024 > span) pos = 0;
        if (MODE == 1 || MODE == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K > 0 ? 2 * K : 0) : "memory");   // two steps of copies in flight
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = (float)x;
    for (int k = 0; k < 8; ++k) for (int g = 0; g < 16; ++g) s += acc[k][g];
    sink[blockIdx.x * 256 + threadIdx.x] = s + (float)lds[threadIdx.x].x;
    if (lane == 0) stamps[blockIdx.x * 4 + wave] = c1 - c0;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

template <int MODE, int K>
static void run(const char *name, const uint32_t *opnd, const char *src, size_t span, int iters, float *sink, unsigned long long *stamps, int cus)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_issue<MODE, K>), dim3(cus), dim3(256), 0, 0, opnd, src, span, iters, sink, stamps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 1 && ms < best) best = ms;
    }
    std::vector<unsigned long long> st((size_t)cus * 4);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    double c = 0; for (auto v : st) c += (double)v;
    const double cyc = c / st.size() / iters;
    printf("%-44s K=%d: %7.0f cycles/step (32 MFMAs = 1024)  %8.3f ms  -> %.2f us/step, %.1f GB/s per CU, %.2f TB/s chip\n", name, K, cyc,
           best, best * 1e3 / iters, 4.0 * K * 1024 / (best * 1e-3 / iters) / 1e9, 4.0 * K * 1024 * cus / (best * 1e-3 / iters) / 1e12);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    uint32_t *opnd; float *sink; unsigned long long *stamps; char *big;
    const size_t span_big = (size_t)16 << 30, span_small = (size_t)16 << 20;
    CK(hipMalloc(&opnd, 6 * 8 * 64 * 4)); CK(hipMalloc(&sink, (size_t)cus * 256 * 4)); CK(hipMalloc(&stamps, (size_t)cus * 4 * 8));
    CK(hipMalloc(&big, span_big)); CK(hipMemset(big, 0x11, span_big));
    std::vector<uint32_t> h(6 * 8 * 64);
    for (auto &w : h) w = rnd() * 2654435761u;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) for (int l = 0; l < 64; ++l) h[(q * 8 + i) * 64 + l] &= 0x33333333u & (rnd() * 2654435761u);
    CK(hipMemcpy(opnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int s = 0; s < 2; ++s) {
        const size_t span = s ? span_big : span_small;
        printf("---- source: %s\n", s ? "16 GB stream (HBM)" : "16 MB buffer (L2 / Infinity Cache)");
        run<0, 0>("no loads", opnd, big, span, iters, sink, stamps, cus);
        run<1, 4>("LDS-DMA burst", opnd, big, span, iters, sink, stamps, cus);
        run<1, 7>("LDS-DMA burst", opnd, big, span, iters, sink, stamps, cus);
        run<2, 4>("LDS-DMA spread", opnd, big, span, iters, sink, stamps, cus);
        run<2, 7>("LDS-DMA spread (K=8 slots)", opnd, big, span, iters, sink, stamps, cus);
        run<3, 4>("global_load_dwordx4 -> VGPR burst", opnd, big, span, iters, sink, stamps, cus);
        run<3, 7>("global_load_dwordx4 -> VGPR burst", opnd, big, span, iters, sink, stamps, cus);
        run<4, 4>("global_load_dwordx4 -> VGPR spread", opnd, big, span, iters, sink, stamps, cus);
        run<4, 8>("global_load_dwordx4 -> VGPR spread", opnd, big, span, iters, sink, stamps, cus);
    }
    return 0;
}
