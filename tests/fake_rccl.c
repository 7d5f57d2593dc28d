/* fake_rccl.c -- TEST-ONLY stand-in for librccl, for ranks that SHARE ONE GPU.
 *
 * The GPU test box has one device and RCCL refuses two ranks on one device, so the N > 1 branches of the library's native
 * communicator (mendeliht.jl_amd/csrc/comm.hip: the all-gather layout, the growth of its staging buffer, the order of its
 * private stream against the fit's stream, the teardown order) never ran.  This file implements the six entry points comm.hip
 * resolves with dlsym -- ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclAllGather / ncclCommDestroy /
 * ncclGetErrorString -- over POSIX shared memory + hipMemcpy, and is loaded through the library's existing override
 * (MENDELIHT_RCCL_LIB).  It includes the REAL <rccl/rccl.h>: the definitions below must match the real prototypes to compile, and
 * the data type / reduction codes comm.hip declares by hand (kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2) are checked here
 * against the header's names -- a wrong code comes back as ncclInvalidArgument.
 *
 * Not a product path: nothing under mendeliht.jl_amd/ links or loads it; tests/ builds it with gcc and points
 * MENDELIHT_RCCL_LIB at it.  Reductions are summed in RANK ORDER on every rank (a real ring sums each chunk in a different
 * order), so every rank holds bit-identical results, which is all the column-sharded fit asks of its communicator.
 *
 * Round 5: the all-reduce of n-vectors runs ON THE DEVICE when it can -- every rank owns a staging buffer in device memory, the
 * ranks open each other's through HIP IPC at set-up, and one small kernel (tests/fake_rccl_kernels.hip, a code object loaded with
 * hipModuleLoad) sums them in rank order into the caller's buffer -- ~40 us instead of the 0.65 ms of two 4 MB copies through host
 * memory and a host loop, so that a two-rank bench line on the one test GPU shows the N > 1 path's own overhead rather than the
 * stand-in's.  Still host-synchronous (two barriers in shared memory around the kernel); if the code object or IPC is not
 * available on any rank, every rank takes the host path (MIH_FAKE_RCCL_HOST=1 forces it).
 *
 * Build: gcc -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.c -o tests/libfake_rccl.so \
 *            -L/opt/rocm/lib -lamdhip64 -lrt -lpthread -ldl
 *        hipcc --genco --offload-arch=gfx950 -O2 tests/fake_rccl_kernels.hip -o tests/fake_rccl_kernels.hsaco   (beside the .so)
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#define FAKE_MAGIC 0x4d49484652434c31ull /* "MIHFRCL1" */

typedef struct {
    _Atomic uint64_t magic;
    _Atomic int32_t attached;          /* ranks that have mapped the segment */
    _Atomic int32_t arrived;           /* sense-reversing barrier */
    _Atomic int32_t generation;
    _Atomic int32_t detached;
    int32_t nranks;
    int64_t slot_bytes;
    _Atomic int64_t calls[4];          /* allreduce, allgather, bytes reduced, bytes gathered: read by the test through fake_rccl_stats */
    _Atomic int32_t dev_fail;          /* ranks that could not set up the device path: > 0 = every rank takes the host path */
    hipIpcMemHandle_t ipc[16];         /* the ranks' device staging buffers */
} fake_header;

typedef struct { const double *p[16]; int n; } fake_peers;            /* = FakePeers of fake_rccl_kernels.hip */

struct ncclComm {                      /* rccl.h: typedef struct ncclComm* ncclComm_t */
    fake_header *hdr;
    char *slots;                       /* nranks slots of slot_bytes behind the header */
    size_t map_bytes;
    int rank, nranks;
    char name[64];
    void *bounce;                      /* host staging of this rank */
    size_t bounce_bytes;
    int dev_ok;                        /* the all-reduce runs on the device */
    void *stage;                       /* this rank's device staging buffer (slot_bytes) */
    void *peer[16];                    /* every rank's staging buffer as this process sees it (peer[rank] = stage) */
    hipModule_t mod; hipFunction_t fn;
};

static const double kTimeoutS = 120.0;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static size_t slot_bytes_default(void)
{
    const char *e = getenv("MIH_FAKE_RCCL_SLOT_MB");
    size_t mb = e ? (size_t)atol(e) : 16;
    if (mb < 1) mb = 1;
    return mb << 20;
}

/* all ranks of the communicator; 0 = ok, -1 = a rank never came (the test fails instead of hanging the box) */
static int fake_barrier(struct ncclComm *c)
{
    fake_header *h = c->hdr;
    const int gen = atomic_load(&h->generation);
    if (atomic_fetch_add(&h->arrived, 1) == c->nranks - 1) {
        atomic_store(&h->arrived, 0);
        atomic_fetch_add(&h->generation, 1);
        return 0;
    }
    const double t0 = now_s();
    for (unsigned spin = 0; atomic_load(&h->generation) == gen; ++spin) {
        if ((spin & 1023u) == 1023u) {
            if (now_s() - t0 > kTimeoutS) return -1;
            usleep(50);
        } else sched_yield();
    }
    return 0;
}

static int ensure_bounce(struct ncclComm *c, size_t bytes)
{
    if (bytes <= c->bounce_bytes) return 0;
    free(c->bounce);
    c->bounce = malloc(bytes);
    c->bounce_bytes = c->bounce ? bytes : 0;
    return c->bounce ? 0 : -1;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId)
{
    static _Atomic int counter = 0;
    if (!uniqueId) return ncclInvalidArgument;
    memset(uniqueId->internal, 0, NCCL_UNIQUE_ID_BYTES);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(uniqueId->internal, NCCL_UNIQUE_ID_BYTES, "/mih_fake_rccl_%d_%d_%lx", (int)getpid(), atomic_fetch_add(&counter, 1),
             (unsigned long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || commId.internal[0] != '/') return ncclInvalidArgument;
    struct ncclComm *c = (struct ncclComm *)calloc(1, sizeof(*c));
    if (!c) return ncclSystemError;
    c->rank = rank; c->nranks = nranks;
    memcpy(c->name, commId.internal, sizeof(c->name) - 1);
    const size_t slot = slot_bytes_default();
    c->map_bytes = 4096 + slot * (size_t)nranks;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { free(c); return ncclSystemError; }
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); free(c); return ncclSystemError; }     /* same size from every rank */
    void *m = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { free(c); return ncclSystemError; }
    c->hdr = (fake_header *)m;
    c->slots = (char *)m + 4096;
    if (rank == 0) {                   /* a fresh segment is zero-filled: rank 0 publishes the geometry, the others wait for it */
        c->hdr->nranks = nranks;
        c->hdr->slot_bytes = (int64_t)slot;
        atomic_store(&c->hdr->magic, FAKE_MAGIC);
    } else {
        const double t0 = now_s();
        while (atomic_load(&c->hdr->magic) != FAKE_MAGIC) {
            if (now_s() - t0 > kTimeoutS) { munmap(m, c->map_bytes); free(c); return ncclSystemError; }
            usleep(100);
        }
        if (c->hdr->nranks != nranks || c->hdr->slot_bytes != (int64_t)slot) { munmap(m, c->map_bytes); free(c); return ncclInvalidArgument; }
    }
    atomic_fetch_add(&c->hdr->attached, 1);
    if (fake_barrier(c) != 0) { munmap(m, c->map_bytes); free(c); return ncclSystemError; }
    if (rank == 0) shm_unlink(c->name);          /* every rank has it mapped: the name can go, the memory lives until the last unmap */
    /* the device path: staging buffer, its IPC handle published, the kernel's code object; then everybody opens everybody's */
    int ok = nranks <= 16 && !getenv("MIH_FAKE_RCCL_HOST");
    if (ok) {
        char path[4096] = {0};
        const char *e = getenv("MIH_FAKE_RCCL_KERNELS");
        Dl_info di;
        if (e) snprintf(path, sizeof(path), "%s", e);
        else if (dladdr((void *)&ncclCommInitRank, &di) && di.dli_fname) {
            snprintf(path, sizeof(path), "%s", di.dli_fname);
            char *slash = strrchr(path, '/');
            snprintf(slash ? slash + 1 : path, sizeof(path) - (size_t)((slash ? slash + 1 : path) - path), "fake_rccl_kernels.hsaco");
        }
        ok = path[0] && hipModuleLoad(&c->mod, path) == hipSuccess && hipModuleGetFunction(&c->fn, c->mod, "fake_allreduce") == hipSuccess;
        if (ok) ok = hipMalloc(&c->stage, slot) == hipSuccess;
        if (ok) ok = hipIpcGetMemHandle(&c->hdr->ipc[rank], c->stage) == hipSuccess;
        if (!ok) (void)hipGetLastError();
    }
    if (!ok) atomic_fetch_add(&c->hdr->dev_fail, 1);
    if (fake_barrier(c) != 0) { munmap(m, c->map_bytes); free(c); return ncclSystemError; }
    if (atomic_load(&c->hdr->dev_fail) == 0) {
        for (int r = 0; r < nranks && ok; ++r) {
            if (r == rank) c->peer[r] = c->stage;
            else ok = hipIpcOpenMemHandle(&c->peer[r], c->hdr->ipc[r], hipIpcMemLazyEnablePeerAccess) == hipSuccess;
        }
        if (!ok) { (void)hipGetLastError(); atomic_fetch_add(&c->hdr->dev_fail, 1); }
    }
    if (fake_barrier(c) != 0) { munmap(m, c->map_bytes); free(c); return ncclSystemError; }
    c->dev_ok = atomic_load(&c->hdr->dev_fail) == 0;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclInvalidArgument;
    if (comm->dev_ok) {
        (void)fake_barrier(comm);                /* nobody closes a buffer a peer's kernel may still read */
        for (int r = 0; r < comm->nranks; ++r) if (r != comm->rank && comm->peer[r]) (void)hipIpcCloseMemHandle(comm->peer[r]);
        (void)fake_barrier(comm);
    }
    if (comm->stage) (void)hipFree(comm->stage);
    if (comm->mod) (void)hipModuleUnload(comm->mod);
    atomic_fetch_add(&comm->hdr->detached, 1);
    munmap((void *)comm->hdr, comm->map_bytes);
    free(comm->bounce);
    free(comm);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->hdr->nranks;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t result)
{
    switch (result) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "fake_rccl: invalid argument (data type / reduction code / size)";
    case ncclSystemError: return "fake_rccl: system error (shared memory, or a rank did not arrive within the timeout)";
    case ncclUnhandledCudaError: return "fake_rccl: HIP error";
    default: return "fake_rccl: error";
    }
}

/* Both collectives are synchronous here: wait for what is queued on the caller's stream in front of the call, stage through
 * the shared segment, write the result, and return with it in place -- a stricter order than the real enqueue, so a caller that
 * is correct against the real library is correct here; the converse is what the caller's own stream synchronisation guards. */
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    if (datatype != ncclFloat64 || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
    const size_t bytes = count * sizeof(double);
    if ((int64_t)bytes > comm->hdr->slot_bytes) return ncclInvalidArgument;          /* MIH_FAKE_RCCL_SLOT_MB */
    if (comm->dev_ok) {                /* on the device: my vector into my staging buffer, everybody's summed in rank order by one kernel */
        if (hipMemcpyAsync(comm->stage, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
        if (fake_barrier(comm) != 0) return ncclSystemError;                          /* every staging buffer holds its rank's vector */
        fake_peers peers; memset(&peers, 0, sizeof(peers));
        for (int r = 0; r < comm->nranks; ++r) peers.p[r] = (const double *)comm->peer[r];
        peers.n = comm->nranks;
        double *out = (double *)recvbuff; unsigned long long cnt = (unsigned long long)count; int is_max = op == ncclMax;
        void *args[4] = {&out, &peers, &cnt, &is_max};
        unsigned grid = (unsigned)((count + 255) / 256); if (grid > 2048u) grid = 2048u; if (grid == 0) grid = 1;
        if (hipModuleLaunchKernel(comm->fn, grid, 1, 1, 256, 1, 1, 0, stream, args, NULL) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
        if (fake_barrier(comm) != 0) return ncclSystemError;                          /* nobody refills a staging buffer a peer's kernel still reads */
        if (comm->rank == 0) { atomic_fetch_add(&comm->hdr->calls[0], 1); atomic_fetch_add(&comm->hdr->calls[2], (int64_t)bytes); }
        return ncclSuccess;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    double *mine = (double *)(comm->slots + (size_t)comm->rank * (size_t)comm->hdr->slot_bytes);
    if (hipMemcpy(mine, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (fake_barrier(comm) != 0) return ncclSystemError;
    if (ensure_bounce(comm, bytes) != 0) return ncclSystemError;
    double *out = (double *)comm->bounce;
    memcpy(out, comm->slots, bytes);                                                  /* rank 0 first, then 1, 2, ... on EVERY rank */
    for (int r = 1; r < comm->nranks; ++r) {
        const double *o = (const double *)(comm->slots + (size_t)r * (size_t)comm->hdr->slot_bytes);
        if (op == ncclSum) for (size_t i = 0; i < count; ++i) out[i] += o[i];
        else for (size_t i = 0; i < count; ++i) out[i] = out[i] < o[i] ? o[i] : out[i];
    }
    if (fake_barrier(comm) != 0) return ncclSystemError;                              /* nobody overwrites a slot another rank still reads */
    if (hipMemcpy(recvbuff, out, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (comm->rank == 0) { atomic_fetch_add(&comm->hdr->calls[0], 1); atomic_fetch_add(&comm->hdr->calls[2], (int64_t)bytes); }
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream)
{
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    if (datatype != ncclFloat64) return ncclInvalidArgument;
    const size_t bytes = sendcount * sizeof(double);
    if ((int64_t)bytes > comm->hdr->slot_bytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    char *mine = comm->slots + (size_t)comm->rank * (size_t)comm->hdr->slot_bytes;
    if (hipMemcpy(mine, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (fake_barrier(comm) != 0) return ncclSystemError;
    if (ensure_bounce(comm, bytes * (size_t)comm->nranks) != 0) return ncclSystemError;
    for (int r = 0; r < comm->nranks; ++r)                                            /* recv[r * count ..) = rank r's send */
        memcpy((char *)comm->bounce + (size_t)r * bytes, comm->slots + (size_t)r * (size_t)comm->hdr->slot_bytes, bytes);
    if (fake_barrier(comm) != 0) return ncclSystemError;
    if (hipMemcpy(recvbuff, comm->bounce, bytes * (size_t)comm->nranks, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (comm->rank == 0) { atomic_fetch_add(&comm->hdr->calls[1], 1); atomic_fetch_add(&comm->hdr->calls[3], (int64_t)(bytes * (size_t)comm->nranks)); }
    return ncclSuccess;
}
