// mfma_probe.hip -- check operand / accumulator lane maps of the MFMA forms considered for the exact
// fixed-point X'r path, with exact small-integer data (stand-alone probe, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// mode 0: A fp4, B fp4; mode 1: A fp4, B fp8(e4m3); mode 2: i8 32x32x32; mode 3: A fp4, B fp6 (e2m3)
__global__ void k_probe(const uint32_t *A, const uint32_t *B, float *D, int mode, int scale)
{
    int l = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)A[l * 8 + i]; b[i] = (int)B[l * 8 + i]; }
    if (mode == 2) {
        i32x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
        i32x16 c = {};
        c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c, 0, 0, 0);
        for (int i = 0; i < 16; ++i) D[l * 16 + i] = (float)c[i];
        return;
    }
    f32x16 c = {};
    if (scale == -1) {      // literal zero scales: the compiler selects the unscaled v_mfma_f32_32x32x64_f8f6f4
        if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
        else if (mode == 3) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 2, 0, 0, 0, 0);
        else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 0, 0, 0, 0, 0);
        for (int i = 0; i < 16; ++i) D[l * 16 + i] = c[i];
        return;
    }
    if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, scale, 0, scale);
    else if (mode == 3) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 2, 0, scale, 0, scale);
    else           c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 0, 0, scale, 0, scale);
    for (int i = 0; i < 16; ++i) D[l * 16 + i] = c[i];
}

static const float FP4V[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static float fp4(int code) { float v = FP4V[code & 7]; return (code & 8) ? -v : v; }
static int e4m3_of_int(int v)   // exact for |v| <= 15
{
    int s = v < 0; int a = abs(v);
    if (a == 0) return s << 7;
    int e = 0; while ((a >> (e + 1)) != 0) ++e;            // a in [2^e, 2^(e+1))
    int mant = ((a << 3) >> e) & 7;                         // 3 mantissa bits
    return (s << 7) | ((e + 7) << 3) | mant;
}

// e2m3 code of u/8 for u in the representable set (0..16, even to 30, multiples of 4 to 60)
static int e2m3_of_units(int v)
{
    int s = v < 0, u = abs(v), e, m;
    if (u < 8) { e = 0; m = u; } else if (u < 16) { e = 1; m = u - 8; } else if (u <= 30) { e = 2; m = u / 2 - 8; } else { e = 3; m = u / 4 - 8; }
    return (s << 5) | (e << 3) | m;
}
static const int E2M3U[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 44, 48, 52, 56, 60};

int main()
{
    uint32_t *dA, *dB; float *dD;
    CK(hipMalloc((void **)&dA, 64 * 8 * 4)); CK(hipMalloc((void **)&dB, 64 * 8 * 4)); CK(hipMalloc((void **)&dD, 64 * 16 * 4));
    srand(7);
    for (int mode = 0; mode < 4; ++mode) {
        const int K = (mode == 2) ? 32 : 64, KH = K / 2;          // K elements per lane-half
        std::vector<double> Am(32 * K), Bm(K * 32);
        std::vector<uint32_t> hA(64 * 8, 0), hB(64 * 8, 0);
        // hypothesis: lane l = (r = l & 31, h = l >> 5) holds A[r][h*KH + j] and B[h*KH + j][r], element j packed little-endian
        for (int l = 0; l < 64; ++l) {
            int r = l & 31, h = l >> 5;
            for (int j = 0; j < KH; ++j) {
                int k = h * KH + j;
                if (mode == 2) {
                    int av = rand() % 3, bv = rand() % 255 - 127;
                    Am[r * K + k] = av; Bm[k * 32 + r] = bv;
                    hA[l * 8 + j / 4] |= (uint32_t)(av & 0xff) << (8 * (j % 4));
                    hB[l * 8 + j / 4] |= (uint32_t)(bv & 0xff) << (8 * (j % 4));
                } else {
                    int ac = rand() % 3;                               // dosage code 0,1,2 -> fp4 0, 0.5, 1.0
                    Am[r * K + k] = fp4(ac);
                    hA[l * 8 + j / 8] |= (uint32_t)ac << (4 * (j % 8));
                    if (mode == 0) {
                        int bc = rand() % 16; if (bc == 8) bc = 0;     // any fp4 value (avoid -0)
                        Bm[k * 32 + r] = fp4(bc);
                        hB[l * 8 + j / 8] |= (uint32_t)bc << (4 * (j % 8));
                    } else if (mode == 3) {
                        int bu = E2M3U[rand() % 32] * ((rand() & 1) ? -1 : 1);
                        Bm[k * 32 + r] = bu / 8.0;
                        uint64_t code = (uint64_t)e2m3_of_units(bu);          // hypothesis: element j in bits 6j .. 6j+5 of the lane's 192 bits
                        int bit = 6 * j, w = bit / 32, sh = bit % 32;
                        hB[l * 8 + w] |= (uint32_t)(code << sh);
                        if (sh > 26) hB[l * 8 + w + 1] |= (uint32_t)(code >> (32 - sh));
                    } else {
                        int bv = rand() % 31 - 15;
                        Bm[k * 32 + r] = bv;
                        hB[l * 8 + j / 4] |= (uint32_t)e4m3_of_int(bv) << (8 * (j % 4));
                    }
                }
            }
        }
        CK(hipMemcpy(dA, hA.data(), 64 * 8 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), 64 * 8 * 4, hipMemcpyHostToDevice));
        for (int scale : {127, 0x7f7f7f7f, 128, -1}) {
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode, scale);
            CK(hipDeviceSynchronize());
            std::vector<float> hD(64 * 16);
            CK(hipMemcpy(hD.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost));
            // C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
            int bad = 0; double maxerr = 0, ratio = 0;
            for (int l = 0; l < 64; ++l) for (int g = 0; g < 16; ++g) {
                int col = l & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
                double ref = 0; for (int k = 0; k < K; ++k) ref += Am[row * K + k] * Bm[k * 32 + col];
                double err = fabs(hD[l * 16 + g] - ref);
                if (err > 1e-6) { if (bad < 3) printf("   mismatch lane %d reg %d: got %g want %g\n", l, g, hD[l * 16 + g], ref); ++bad; if (ref != 0) ratio = hD[l * 16 + g] / ref; }
                if (err > maxerr) maxerr = err;
            }
            printf("mode %d (%s) scale=0x%x: mismatches %d / 1024, max err %g%s\n", mode,
                   mode == 0 ? "fp4 x fp4" : mode == 1 ? "fp4 x fp8" : mode == 3 ? "fp4 x fp6" : "i8 x i8", scale, bad, maxerr, bad ? "  <-- natural hypothesis fails" : "  OK");
            if (bad) printf("   last got/want ratio %g\n", ratio);
            if (mode == 2) break;
        }
    }
    return 0;
}
