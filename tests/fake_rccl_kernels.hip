// fake_rccl_kernels.hip -- the one kernel of the TEST-ONLY librccl stand-in (tests/fake_rccl.c): an all-reduce over the ranks' staging
// buffers, which the ranks -- processes sharing ONE GPU -- have opened in each other through HIP IPC.  Built as a code object
// (hipcc --genco --offload-arch=gfx950 tests/fake_rccl_kernels.hip -o tests/fake_rccl_kernels.hsaco) and loaded by the C file with
// hipModuleLoad, so that the stand-in itself stays plain C compiled against the real <rccl/rccl.h>.
// Summed in RANK ORDER on every rank, as the host path of the stand-in does: every rank holds bit-identical results.
#include <hip/hip_runtime.h>
struct FakePeers { const double *p[16]; int n; };
extern "C" __global__ void __launch_bounds__(256) fake_allreduce(double *out, FakePeers peers, unsigned long long count, int is_max)
{
    const unsigned long long stride = 256ull * gridDim.x;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < count; i += stride) {
        double a = peers.p[0][i];
        for (int r = 1; r < peers.n; ++r) {
            const double o = peers.p[r][i];
            a = is_max ? (a < o ? o : a) : a + o;
        }
        out[i] = a;
    }
}
