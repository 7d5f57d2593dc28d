// What feeds PCIe fastest from the caller's pageable .bed columns?  (1) hipHostRegister of the caller's memory in place, whole and
// in chunks, and the H2D rate out of it; (2) the multi-threaded copy into a pinned staging buffer (the round-2 pipeline), by thread
// count; (3) hipMemcpy straight from pageable memory.  Build: hipcc --offload-arch=gfx950 -O3 tools/ingest_probe.hip -o build/ingest_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const size_t GB = 1ull << 30, total = (argc > 1 ? atoll(argv[1]) : 8) * GB, chunk = 256ull << 20;
    char *src = (char *)aligned_alloc(4096, total);
    for (size_t i = 0; i < total; i += 4096) src[i] = (char)i;          // resident pageable memory
    char *dev; CK(hipMalloc((void **)&dev, total));
    hipStream_t s; CK(hipStreamCreate(&s));
    // pinned reference
    char *pin; CK(hipHostMalloc((void **)&pin, chunk, hipHostMallocDefault));
    memset(pin, 1, chunk);
    { double t0 = now(); for (int r = 0; r < 16; ++r) CK(hipMemcpyAsync(dev + r * chunk, pin, chunk, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
      printf("pinned staging buffer -> device: %.1f GB/s\n", 16 * chunk / (now() - t0) / 1e9); }
    // (1) register whole
    { double t0 = now(); CK(hipHostRegister(src, total, hipHostRegisterDefault)); double t1 = now();
      CK(hipMemcpyAsync(dev, src, total, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now();
      CK(hipHostUnregister(src)); double t3 = now();
      printf("register %zu GB in place: %.3f s (%.1f GB/s), H2D out of it %.1f GB/s, unregister %.3f s; end to end %.1f GB/s\n", total / GB, t1 - t0,
             total / (t1 - t0) / 1e9, total / (t2 - t1) / 1e9, t3 - t2, total / (t3 - t0) / 1e9); }
    // (1b) chunked register, pipelined by a helper thread two chunks ahead
    for (size_t ck : {256ull << 20, 1ull << 30}) {
        const size_t nchunks = total / ck;
        double t0 = now();
        std::vector<int> ready(nchunks, 0);
        std::thread reg([&]() { for (size_t c = 0; c < nchunks; ++c) { (void)hipHostRegister(src + c * ck, ck, hipHostRegisterDefault); __atomic_store_n(&ready[c], 1, __ATOMIC_RELEASE); } });
        for (size_t c = 0; c < nchunks; ++c) {
            while (!__atomic_load_n(&ready[c], __ATOMIC_ACQUIRE)) std::this_thread::yield();
            CK(hipMemcpyAsync(dev + c * ck, src + c * ck, ck, hipMemcpyHostToDevice, s));
        }
        CK(hipStreamSynchronize(s));
        reg.join();
        double t1 = now();
        for (size_t c = 0; c < nchunks; ++c) (void)hipHostUnregister(src + c * ck);
        double t2 = now();
        printf("chunked register (%zu MB, helper thread) + H2D: %.1f GB/s (+ unregister %.3f s => %.1f GB/s)\n", ck >> 20, total / (t1 - t0) / 1e9, t2 - t1, total / (t2 - t0) / 1e9);
    }
    // (2) threaded copy into pinned staging, double buffered
    char *pin2; CK(hipHostMalloc((void **)&pin2, chunk, hipHostMallocDefault));
    for (int nth : {4, 8, 12, 16, 24, 32}) {
        char *pb[2] = {pin, pin2}; hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
        double t0 = now();
        const size_t nchunks = total / chunk;
        for (size_t c = 0; c < nchunks; ++c) {
            const int b = c & 1;
            if (c >= 2) CK(hipEventSynchronize(ev[b]));
            std::vector<std::thread> th; const size_t per = chunk / nth;
            for (int t = 0; t < nth; ++t) th.emplace_back([=]() { memcpy(pb[b] + t * per, src + c * chunk + t * per, per); });
            for (auto &t : th) t.join();
            CK(hipMemcpyAsync(dev + c * chunk, pb[b], chunk, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[b], s));
        }
        CK(hipStreamSynchronize(s));
        printf("copy with %2d threads into pinned staging + H2D: %.1f GB/s\n", nth, total / (now() - t0) / 1e9);
    }
    // (3) pageable
    { double t0 = now(); CK(hipMemcpy(dev, src, total / 4, hipMemcpyHostToDevice)); printf("hipMemcpy from pageable memory: %.1f GB/s\n", total / 4 / (now() - t0) / 1e9); }
    return 0;
}
