// comm.hip -- a native exchange for the column-sharded fit (second row): an RCCL communicator owned by the
// library.  The fit itself only knows `mih_comm` = {rank, world, column range, allreduce, allgather, user}
// (include/mendeliht_hip.h); a host language may implement the two callbacks with its own communicator (torch.distributed
// in the Python mirror, MPI in the Julia glue) -- or ask the library for this one, whose callbacks run ncclAllReduce /
// ncclAllGather on a private stream over xGMI without re-entering the host language.  One process per GPU; the 128-byte
// unique id travels from rank 0 to the others through whatever the launcher has (MPI_Bcast, torch.distributed, a file).
// librccl is loaded lazily (dlopen) so that single-GPU users never need it.
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

namespace mih {

// the handful of RCCL entry points used, resolved at first use (types as in <rccl/rccl.h>, which is not included so that
// the library builds without it)
typedef struct { char internal[128]; } nccl_uid;
typedef void *nccl_comm;
enum { kNcclSuccess = 0, kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2 };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm *, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(nccl_comm, int *) = nullptr;      // optional (diagnostics: mih_comm_info)
    std::string path;                                   // what dlopen was given
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);          // two lanes / host threads may make their first call together
    if (g_rccl.lib) return MIH_OK;
    // RCCL must drive the SAME HIP runtime this library is bound to.  A host process may hold two (PyTorch ships its own copies of
    // libamdhip64 / libhsa-runtime64 / librccl next to the system's): asked for by its soname, dlopen hands back whichever librccl is
    // loaded already -- possibly the one bound to the OTHER runtime, which then finds "no ROCm-capable device" because that runtime
    // was never initialised (seen when this library is loaded before `import torch`).  So the first candidates are the librccl in
    // the directory of the libamdhip64 that hipGetDeviceCount resolves to, by full path.
    std::string beside[2];
    Dl_info info;
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) { dir.resize(slash + 1); beside[0] = dir + "librccl.so.1"; beside[1] = dir + "librccl.so"; }
    }
    const char *names[] = {getenv("MENDELIHT_RCCL_LIB"), beside[0].empty() ? nullptr : beside[0].c_str(),
                           beside[1].empty() ? nullptr : beside[1].c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    const char *opened = nullptr;
    for (const char *nm : names) { if (nm && (lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) { opened = nm; break; } }
    if (!lib) { set_error("cannot load librccl (%s): set MENDELIHT_RCCL_LIB", dlerror()); return MIH_BAD_ARG; }
    Rccl r;
    r.lib = lib;
    r.GetUniqueId = (int (*)(nccl_uid *))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(nccl_comm *, int, nccl_uid, int))dlsym(lib, "ncclCommInitRank");
    r.CommDestroy = (int (*)(nccl_comm))dlsym(lib, "ncclCommDestroy");
    r.AllReduce = (int (*)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t))dlsym(lib, "ncclAllReduce");
    r.AllGather = (int (*)(const void *, void *, size_t, int, nccl_comm, hipStream_t))dlsym(lib, "ncclAllGather");
    r.GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
    r.CommCount = (int (*)(nccl_comm, int *))dlsym(lib, "ncclCommCount");
    r.path = opened ? opened : "";
    { Dl_info li; if (r.GetUniqueId && dladdr((void *)r.GetUniqueId, &li) && li.dli_fname) r.path = li.dli_fname; }      // the file that answered
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.AllGather) {
        set_error("librccl lacks an expected entry point");
        dlclose(lib);
        return MIH_BAD_ARG;
    }
    g_rccl = r;
    return MIH_OK;
}

struct NativeComm {
    mih_comm c;                 // first member: the handle IS a mih_comm
    nccl_comm comm = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf<double> stage;       // device staging of the host-side exchanges (a few doubles .. world * K)
    size_t stage_cap = 0;
};

static int nccl_fail(int rc, const char *what)
{
    set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
    return MIH_HIP_ERROR;
}

static int ensure_stage(NativeComm *nc, size_t doubles)
{
    if (doubles <= nc->stage_cap) return MIH_OK;
    //  the callbacks run inside mih_fit_iht, whose pool / arena scopes are active: the communicator may outlive the
    // matrix, so its staging buffer must not come out of the matrix's reserve
    PoolScope no_pool(nullptr); ArenaScope no_arena(nullptr);
    MIH_TRY(nc->stage.alloc(doubles * 2));
    nc->stage_cap = doubles * 2;
    return MIH_OK;
}

// mih_comm::allreduce -- in-place sum / max over the ranks.  on_device: buf is device memory of the fit's GPU and the fit's
// stream has been synchronised by the caller; the reduced values are visible to every stream when this returns.
static int native_allreduce(void *user, double *buf, int64_t count, int32_t op, int32_t on_device)
{
    NativeComm *nc = static_cast<NativeComm *>(user);
    if (hipSetDevice(nc->device) != hipSuccess) return 1;
    const int rop = op == 0 ? kNcclSum : kNcclMax;
    if (on_device) {
        if (g_rccl.AllReduce(buf, buf, (size_t)count, kNcclFloat64, rop, nc->comm, nc->stream) != kNcclSuccess) return 2;
        return hipStreamSynchronize(nc->stream) == hipSuccess ? 0 : 3;
    }
    if (ensure_stage(nc, (size_t)count)) return 4;
    if (hipMemcpyAsync(nc->stage.p, buf, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, nc->stream) != hipSuccess) return 5;
    if (g_rccl.AllReduce(nc->stage.p, nc->stage.p, (size_t)count, kNcclFloat64, rop, nc->comm, nc->stream) != kNcclSuccess) return 2;
    if (hipMemcpyAsync(buf, nc->stage.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, nc->stream) != hipSuccess) return 5;
    return hipStreamSynchronize(nc->stream) == hipSuccess ? 0 : 3;
}

// mih_comm::allgather -- recv[r*count .. (r+1)*count) = `send` of rank r; host memory on both sides
static int native_allgather(void *user, const double *send, int64_t count, double *recv)
{
    NativeComm *nc = static_cast<NativeComm *>(user);
    if (hipSetDevice(nc->device) != hipSuccess) return 1;
    const size_t w = (size_t)nc->c.world, cnt = (size_t)count;
    if (ensure_stage(nc, cnt * (w + 1))) return 4;
    double *s = nc->stage.p, *r = nc->stage.p + cnt;
    if (hipMemcpyAsync(s, send, sizeof(double) * cnt, hipMemcpyHostToDevice, nc->stream) != hipSuccess) return 5;
    if (g_rccl.AllGather(s, r, cnt, kNcclFloat64, nc->comm, nc->stream) != kNcclSuccess) return 2;
    if (hipMemcpyAsync(recv, r, sizeof(double) * cnt * w, hipMemcpyDeviceToHost, nc->stream) != hipSuccess) return 5;
    return hipStreamSynchronize(nc->stream) == hipSuccess ? 0 : 3;
}

// The n-vector sums of a column-sharded fit (X_S b_S, X_S g_S) when the communicator is the library's own: ncclAllReduce is queued
// on the FIT's stream, behind the kernel that produced the partial sums and in front of the kernels that read the totals --
// no host synchronisation on either side (the callback contract of mih_comm::allreduce needs two).  -1: not a native communicator,
// or the fit's stream lives on another device than the communicator: the caller then takes the callback route, which
// synchronises the fit's stream and runs the collective on the communicator's own stream.
// INVARIANT the caller keeps: a collective queued here on the fit's stream is followed by a HOST WAIT on that stream before the next
// collective of this communicator is issued from any other stream (the all-gather of project_full_sharded runs on nc->stream):
// every such all-reduce sits in front of a final_sum_home / hipStreamSynchronize of the same chain (IhtVar::update_xb -> mu_loglik,
// stepsize / step_post_fused -> final_sum_home), so two collectives of one communicator are never in flight on two streams.
int comm_native_allreduce_on_stream(const mih_comm *c, double *buf_dev, int64_t count, int32_t op, hipStream_t s, int device)
{
    if (!c || c->allreduce != native_allreduce || c->user != (void *)c) return -1;
    NativeComm *nc = static_cast<NativeComm *>(c->user);
    if (device != nc->device) return -1;
    const int rc = g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, kNcclFloat64, op == 0 ? kNcclSum : kNcclMax, nc->comm, s);
    if (rc != kNcclSuccess) return nccl_fail(rc, "ncclAllReduce");
    return MIH_OK;
}

// ... and the all-gather of a device-resident sharded step (every shard's top-K candidates of project_k!): device buffers on both
// sides, queued on the fit's stream like the all-reduces above (same invariant: the chain that follows reads the result in stream
// order, and the host issues the collectives of every rank in the same order).  -1: not the library's communicator / another device.
int comm_native_allgather_on_stream(const mih_comm *c, const double *send_dev, double *recv_dev, int64_t count, hipStream_t s, int device)
{
    if (!c || c->allreduce != native_allreduce || c->user != (void *)c) return -1;
    NativeComm *nc = static_cast<NativeComm *>(c->user);
    if (device != nc->device) return -1;
    const int rc = g_rccl.AllGather(send_dev, recv_dev, (size_t)count, kNcclFloat64, nc->comm, s);
    if (rc != kNcclSuccess) return nccl_fail(rc, "ncclAllGather");
    return MIH_OK;
}
bool comm_is_native(const mih_comm *c, int device)
{
    return c && c->allreduce == native_allreduce && c->user == (void *)c && static_cast<NativeComm *>(c->user)->device == device;
}

}  // namespace mih

using namespace mih;

extern "C" {

int mih_rccl_unique_id(void *id128)
{
    if (!id128) { set_error("null argument"); return MIH_BAD_ARG; }
    MIH_TRY(rccl_load());
    nccl_uid id;
    int rc = g_rccl.GetUniqueId(&id);
    if (rc != kNcclSuccess) return nccl_fail(rc, "ncclGetUniqueId");
    std::memcpy(id128, id.internal, 128);
    return MIH_OK;
}

int mih_comm_create_rccl(const void *id128, int32_t rank, int32_t world, int32_t device, int64_t col_offset,
                         int64_t p_global, mih_comm **out)
{
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world || col_offset < 0 || p_global < 1) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); set_error("no HIP device available"); return MIH_NO_DEVICE; }
    MIH_TRY(rccl_load());
    MIH_HIP(hipSetDevice(device));
    NativeComm *nc = new NativeComm();
    nc->device = device;
    nccl_uid id;
    std::memcpy(id.internal, id128, 128);
    int rc = g_rccl.CommInitRank(&nc->comm, world, id, rank);
    if (rc != kNcclSuccess) { delete nc; return nccl_fail(rc, "ncclCommInitRank"); }
    if (hipStreamCreate(&nc->stream) != hipSuccess) { g_rccl.CommDestroy(nc->comm); delete nc; set_error("hipStreamCreate failed"); return MIH_HIP_ERROR; }
    nc->c.rank = rank; nc->c.world = world; nc->c.col_offset = col_offset; nc->c.p_global = p_global;
    nc->c.allreduce = native_allreduce; nc->c.allgather = native_allgather; nc->c.user = nc;
    *out = &nc->c;
    return MIH_OK;
}

// The single exchange of a cross-validation run by one process per GPU: every rank holds the losses of its own (fold, k)
// combinations (zeros elsewhere, mih_cv_iht); one all-gather, summed in rank order on every rank.
int mih_cv_allgather(const mih_comm *c, double *mses_raw, int64_t count)
{
    if (!c || !mses_raw || count < 0 || !c->allgather || c->world < 1) { set_error("null/invalid argument"); return MIH_BAD_ARG; }
    if (c->world == 1 || count == 0) return MIH_OK;
    std::vector<double> all((size_t)count * (size_t)c->world);
    const int rc = c->allgather(c->user, mses_raw, count, all.data());
    if (rc) { set_error("communicator all-gather failed (%d)", rc); return MIH_HIP_ERROR; }
    for (int64_t i = 0; i < count; ++i) {
        double s = 0.0;
        for (int32_t r = 0; r < c->world; ++r) s += all[(size_t)r * (size_t)count + (size_t)i];
        mses_raw[i] = s;
    }
    return MIH_OK;
}

// diagnostics of the library's own communicator: the rank count RCCL itself reports (ncclCommCount; -1 if the library lacks it) and
// the librccl file that was loaded (dladdr of its ncclGetUniqueId)
int mih_comm_info(const mih_comm *c, int32_t *ranks_seen, char *librccl_path, int64_t cap)
{
    if (!c || c->allreduce != native_allreduce || c->user != (void *)c) { set_error("not a communicator made by mih_comm_create_rccl"); return MIH_BAD_ARG; }
    NativeComm *nc = static_cast<NativeComm *>(c->user);
    if (ranks_seen) {
        int cnt = -1;
        if (g_rccl.CommCount && g_rccl.CommCount(nc->comm, &cnt) != kNcclSuccess) cnt = -1;
        *ranks_seen = cnt;
    }
    if (librccl_path && cap > 0) { snprintf(librccl_path, (size_t)cap, "%s", g_rccl.path.c_str()); }
    return MIH_OK;
}

int mih_comm_destroy_rccl(mih_comm *c)
{
    if (!c) return MIH_OK;
    if (c->allreduce != native_allreduce || c->user != (void *)c) { set_error("not a communicator made by mih_comm_create_rccl"); return MIH_BAD_ARG; }
    NativeComm *nc = static_cast<NativeComm *>(c->user);
    (void)hipSetDevice(nc->device);
    if (nc->stream) { (void)hipStreamSynchronize(nc->stream); (void)hipStreamDestroy(nc->stream); }
    if (nc->comm) (void)g_rccl.CommDestroy(nc->comm);
    delete nc;
    return MIH_OK;
}

}  // extern "C"
