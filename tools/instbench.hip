// instbench.hip -- VALU / MFMA issue cost on MI355X in shader cycles per wave64 instruction per SIMD.
// Stand-alone probe (not part of the product).  Each kernel runs ITER x 16 independent copies of one
// instruction; W waves per SIMD run it concurrently; cycles come from s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 2000;

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

#define KERNEL32(NAME, ASMLINE)                                                                       \
__global__ void __launch_bounds__(256) NAME(unsigned long long *t, uint32_t *sink, uint32_t seed)      \
{                                                                                                      \
    uint32_t a[16], b = seed ^ threadIdx.x, c = seed * 3u + threadIdx.x;                               \
    for (int i = 0; i < 16; ++i) a[i] = seed + i * 77u + threadIdx.x;                                  \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int it = 0; it < ITER; ++it) {                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASMLINE : "+v"(a[i]) : "v"(b), "v"(c)); \
    }                                                                                                  \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= a[i];                                            \
    if (s == 0x1234567u) sink[0] = s;                                                                  \
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                     \
}

#define KERNEL64(NAME, ASMLINE)                                                                       \
__global__ void __launch_bounds__(256) NAME(unsigned long long *t, uint32_t *sink, uint32_t seed)      \
{                                                                                                      \
    double a[16], b = 1.0 + 1e-9 * (seed ^ threadIdx.x), c = 1e-9 * threadIdx.x;                       \
    for (int i = 0; i < 16; ++i) a[i] = 1.0 + 1e-3 * i;                                                \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int it = 0; it < ITER; ++it) {                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASMLINE : "+v"(a[i]) : "v"(b), "v"(c)); \
    }                                                                                                  \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];                                              \
    if (s == 0.1234567) sink[0] = 1;                                                                   \
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                     \
}

KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_and_sdwa, "v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL32(k_mov_sdwa, "v_mov_b32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1")
KERNEL32(k_lshl, "v_lshlrev_b32 %0, 2, %0")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %2")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 2, 2")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
KERNEL32(k_dot4_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
KERNEL32(k_dot8_i4, "v_dot8_i32_i4 %0, %1, %2, %0")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %1")

// mixed: the production decode pattern (1 SDWA + 1 f64 FMA per dosage)
__global__ void __launch_bounds__(256) k_mix_sdwa_fma(unsigned long long *t, uint32_t *sink, uint32_t seed)
{
    double acc[8]; for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    double d[4] = {2.0, 2.0, 2.0, 2.0};
    double r = 1.0 + 1e-9 * threadIdx.x;
    uint32_t w = seed ^ (threadIdx.x * 2654435761u), mask = 0x0c;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t h = (uint32_t)__double2hiint(d[i & 3]);
            asm volatile("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(h) : "v"(w), "v"(mask));
            d[i & 3] = __hiloint2double((int)h, __double2loint(d[i & 3]));
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(d[i & 3]), "v"(r));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0.1234567) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// u32 -> f64 conversion (the X_S v decode of round 5 asks: is it a full-rate instruction?)
__global__ void __launch_bounds__(256) k_cvt_f64_u32(unsigned long long *t, uint32_t *sink, uint32_t seed)
{
    double a[16]; uint32_t b = seed ^ threadIdx.x;
    for (int i = 0; i < 16; ++i) a[i] = 0.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 0.1234567) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// the two decodes of a 2-bit dosage for an f64 multiply-add, 16 dosages of one dword per iteration:
//   A: v_bfe_u32, v_cvt_f64_u32, v_fma_f64        B: v_bfe_u32 into the low half of (2^52 | g), v_add_f64 (- 2^52), v_fma_f64
template <int VARIANT>
__global__ void __launch_bounds__(256) k_decode_fma(unsigned long long *t, uint32_t *sink, uint32_t seed)
{
    double acc[16]; for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    const double r = 1.0 + 1e-9 * threadIdx.x;
    uint32_t w = seed ^ (threadIdx.x * 2654435761u);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t g = (w >> (2 * i)) & 3u;
            double d;
            if (VARIANT == 0) d = (double)g;
            else d = __hiloint2double(0x43300000, (int)g) - 4503599627370496.0;
            acc[i] = fma(d, r, acc[i]);
        }
        asm volatile("" : "+v"(w));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 0.1234567) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define MFMA_KERNEL(NAME, ACC_T, DECL, CALL)                                                          \
__global__ void __launch_bounds__(256) NAME(unsigned long long *t, uint32_t *sink, uint32_t seed)      \
{                                                                                                      \
    ACC_T acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};                                                  \
    DECL                                                                                               \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int it = 0; it < ITER; ++it) {                                                                \
        acc0 = CALL(acc0); acc1 = CALL(acc1); acc2 = CALL(acc2); acc3 = CALL(acc3);                    \
        acc0 = CALL(acc0); acc1 = CALL(acc1); acc2 = CALL(acc2); acc3 = CALL(acc3);                    \
        acc0 = CALL(acc0); acc1 = CALL(acc1); acc2 = CALL(acc2); acc3 = CALL(acc3);                    \
        acc0 = CALL(acc0); acc1 = CALL(acc1); acc2 = CALL(acc2); acc3 = CALL(acc3);                    \
    }                                                                                                  \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
    if ((float)acc0[0] + (float)acc1[1] + (float)acc2[2] + (float)acc3[3] == 0.1234567f) sink[0] = 1;  \
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                     \
}

#define I8_DECL i32x4 a = {(int)seed, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
#define I8_CALL(C) __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, C, 0, 0, 0)
MFMA_KERNEL(k_mfma_i8_32, i32x16, I8_DECL, I8_CALL)
#define I8B_CALL(C) __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, C, 0, 0, 0)
MFMA_KERNEL(k_mfma_i8_16, i32x4, I8_DECL, I8B_CALL)
#define F64_DECL double a = 1.0 + seed * 1e-9, b = 1.0 + threadIdx.x * 1e-9;
#define F64_CALL(C) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C, 0, 0, 0)
MFMA_KERNEL(k_mfma_f64_16, f64x4, F64_DECL, F64_CALL)
#define F64B_CALL(C) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, C, 0, 0, 0)
__global__ void __launch_bounds__(256) k_mfma_f64_4(unsigned long long *t, uint32_t *sink, uint32_t seed)
{
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    F64_DECL
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) { acc0 = F64B_CALL(acc0); acc1 = F64B_CALL(acc1); acc2 = F64B_CALL(acc2); acc3 = F64B_CALL(acc3); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc0 + acc1 + acc2 + acc3 == 0.1234567) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// block-scaled f8f6f4: cbsz/blgp select the A/B formats (0 fp8 e4m3, 1 bf8, 2 fp6, 3 bf6, 4 fp4)
#define SC_DECL i32x8 a = {(int)seed, 1, 2, 3, 4, 5, 6, 7}, b = {4, 5, (int)threadIdx.x, 7, 1, 2, 3, 4};
#define SC_FP4FP8(C) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, C, 4, 0, 0, 127, 0, 127)
#define SC_FP4FP4(C) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, C, 4, 4, 0, 127, 0, 127)
#define SC_FP8FP8(C) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, C, 0, 0, 0, 127, 0, 127)
MFMA_KERNEL(k_mfma_sc_fp4_fp8, f32x16, SC_DECL, SC_FP4FP8)
MFMA_KERNEL(k_mfma_sc_fp4_fp4, f32x16, SC_DECL, SC_FP4FP4)
MFMA_KERNEL(k_mfma_sc_fp8_fp8, f32x16, SC_DECL, SC_FP8FP8)
#define SC16_FP4FP8(C) __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, C, 4, 0, 0, 127, 0, 127)
MFMA_KERNEL(k_mfma_sc16_fp4_fp8, f32x4, SC_DECL, SC16_FP4FP8)

static unsigned long long *d_t; static uint32_t *d_sink;

template <typename K> static void run(const char *name, K kern, int waves_per_simd, int inst_per_iter)
{
    int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves = 1 per SIMD
    CK(hipMemset(d_t, 0, sizeof(unsigned long long) * blocks * 4));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_t, d_sink, 12345u);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_t, d_sink, 12345u);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    double per_wave = med / ((double)ITER * inst_per_iter);          // cycles per instruction as one wave sees it
    // wall-clock view: the same launch timed with HIP events (includes launch ramp; ITER is large)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_t, d_sink, 12345u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double ns_per_inst_simd = (double)ms * 1e6 / ((double)ITER * inst_per_iter * waves_per_simd);
    printf("%-22s w/SIMD=%d  ticks/inst(one wave)=%7.2f  ticks/inst/SIMD=%6.2f   wall: %6.3f ns/inst/SIMD  (ticks/us=%.0f)\n",
           name, waves_per_simd, per_wave, per_wave / waves_per_simd, ns_per_inst_simd, med / (ms * 1e3));
    fflush(stdout);
}

int main()
{
    CK(hipMalloc((void **)&d_t, sizeof(unsigned long long) * 256 * 8 * 4)); CK(hipMalloc((void **)&d_sink, 64));
#define R(K, N) run(#K, K, 2, N); run(#K, K, 4, N); run(#K, K, 8, N);
    R(k_and, 16) R(k_and_sdwa, 16) R(k_mov_sdwa, 16) R(k_lshl, 16) R(k_perm, 16) R(k_bfi, 16) R(k_and_or, 16) R(k_lshl_or, 16) R(k_bfe, 16)
    R(k_add_u32, 16) R(k_mad_u24, 16) R(k_mul_lo, 16) R(k_fma_f32, 16) R(k_pk_fma_f16, 16) R(k_dot4_i8, 16) R(k_dot8_i4, 16) R(k_cndmask, 16) R(k_cvt_f32_u32, 16)
    R(k_fma_f64, 16) R(k_add_f64, 16) R(k_mul_f64, 16) R(k_pk_fma_f32, 16) R(k_pk_add_f32, 16) R(k_lshl_add_u64, 16) R(k_mov_b64, 16)
    R(k_mix_sdwa_fma, 32)
    R(k_cvt_f64_u32, 16) R(k_decode_fma<0>, 16) R(k_decode_fma<1>, 16)
    R(k_mfma_i8_32, 16) R(k_mfma_i8_16, 16) R(k_mfma_f64_16, 16) R(k_mfma_f64_4, 16)
    R(k_mfma_sc_fp4_fp8, 16) R(k_mfma_sc_fp4_fp4, 16) R(k_mfma_sc_fp8_fp8, 16) R(k_mfma_sc16_fp4_fp8, 16)
    return 0;
}
