// How long do small pinned / device allocations take?  (cv_iht creates ~4 pinned and ~25 device buffers per IHTVariable.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main()
{
    hipFree(0);
    for (int rep = 0; rep < 3; ++rep) {
        for (size_t bytes : {4096ul, 65536ul, 1ul << 20}) {
            for (int coherent = 0; coherent < 2; ++coherent) {
                std::vector<void *> p(64);
                auto t0 = std::chrono::steady_clock::now();
                for (auto &q : p) hipHostMalloc(&q, bytes, coherent ? hipHostMallocCoherent : hipHostMallocDefault);
                auto t1 = std::chrono::steady_clock::now();
                for (auto &q : p) hipHostFree(q);
                auto t2 = std::chrono::steady_clock::now();
                printf("rep %d hipHostMalloc %8zu B %s: %7.1f us each, hipHostFree %7.1f us each\n", rep, bytes, coherent ? "coherent" : "default ",
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / 64, std::chrono::duration<double, std::micro>(t2 - t1).count() / 64);
            }
        }
        for (size_t bytes : {4096ul, 4ul << 20, 64ul << 20}) {
            std::vector<void *> p(32);
            auto t0 = std::chrono::steady_clock::now();
            for (auto &q : p) hipMalloc(&q, bytes);
            auto t1 = std::chrono::steady_clock::now();
            for (auto &q : p) hipFree(q);
            auto t2 = std::chrono::steady_clock::now();
            printf("rep %d hipMalloc     %8zu B         : %7.1f us each, hipFree     %7.1f us each\n", rep, bytes,
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / 32, std::chrono::duration<double, std::micro>(t2 - t1).count() / 32);
        }
    }
    return 0;
}
