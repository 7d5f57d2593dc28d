// tools/cu_mask_map.hip -- which bit of a stream's CU mask (hipExtStreamCreateWithCUMask) is which compute unit.  For every bit b a
// stream with every bit set EXCEPT b runs a kernel with enough long workgroups to occupy every CU it may use; the waves report
// where they ran (XCC_ID; SE / SH / CU of HW_ID) and the CU that stayed empty is bit b's.  (A mask that leaves an XCD without
// any CU is not honoured -- the XCD gets all of its CUs back -- so single-bit masks tell nothing.)  The lanes' CU reservation
// (csrc/fit_lockstep.hip, lane_stream_create) must take its CUs evenly from the eight XCDs: a kernel's workgroups are dealt to
// the XCDs round-robin, so an XCD that lost more CUs than the others is the straggler of every pass.
//   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_map.hip -o build/cu_mask_map && build/cu_mask_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void k_where(unsigned *out, long long spin)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | ((hw >> 8) & 0xff);      // xcc | se(3) sh(1) cu(4)
}

static std::set<unsigned> run(const std::vector<uint32_t> &mask, unsigned *d, std::vector<unsigned> &h, int nblk)
{
    hipStream_t s;
    std::set<unsigned> seen;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); return seen; }
    (void)hipMemsetAsync(d, 0xff, sizeof(unsigned) * nblk, s);
    hipLaunchKernelGGL(k_where, dim3(nblk), dim3(256), 0, s, d, 3000LL);          // 3000 ticks of the 100 MHz wall clock = 30 us
    (void)hipMemcpyAsync(h.data(), d, sizeof(unsigned) * nblk, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    for (int i = 0; i < nblk; ++i) seen.insert(h[i]);
    return seen;
}

int main()
{
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = pr.multiProcessorCount, words = (cus + 31) / 32, nblk = 16384;
    unsigned *d; (void)hipMalloc(&d, sizeof(unsigned) * nblk);
    std::vector<unsigned> h(nblk);
    std::vector<uint32_t> full(words, 0xffffffffu);
    const std::set<unsigned> all = run(full, d, h, nblk);
    printf("{\"cus\": %d, \"seen_with_full_mask\": %zu, \"bits\": [", cus, all.size());
    for (int b = 0; b < cus; ++b) {
        std::vector<uint32_t> mask = full;
        mask[b >> 5] &= ~(1u << (b & 31));
        const std::set<unsigned> seen = run(mask, d, h, nblk);
        printf("%s{\"bit\": %d, \"seen\": %zu, \"missing\": [", b ? ", " : "", b, seen.size());
        bool first = true;
        for (unsigned v : all) if (!seen.count(v)) { printf("%s[%u, %u, %u, %u]", first ? "" : ", ", v >> 16, (v >> 5) & 7, (v >> 4) & 1, v & 15); first = false; }
        printf("]}");
    }
    printf("]}\n");
    return 0;
}
