"""Before a multi-GPU bench: what does an exchange of the column-sharded fit cost on THIS node, on both communicators?

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/scale_preflight.py

Times, per rank and as the max over ranks: all-reduces of n + 1 doubles (n = 500 000: the 4 MB sum of X_S g_S / X_S b_S) and the
all-gather of 100 doubles (the q x |path| loss matrix of cv_iht), over (a) torch.distributed (backend nccl = RCCL) and (b) the
library's own communicator (mih_comm_create_rccl: csrc/comm.hip, the one bench.py --gpus N uses), and prints ONE JSON line on rank
0 -- with the librccl file the library loaded and the rank count RCCL reports for its communicator.  No re-exec: children only
(the launcher starts the ranks before anything touches a GPU).  MIH_BENCH_BACKEND=gloo MIH_BENCH_ONE_DEVICE=1 MENDELIHT_RCCL_LIB=
tests/libfake_rccl.so runs it on a one-GPU box through the stand-in (a functional check, not a measurement)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("MIH_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MIH_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if world > 1:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    import mendeliht_amd as m
    from mendeliht_amd import api, dist as D
    n, reps = 500_000, 50
    dev = f"cuda:{local}"
    out = {"world": world, "backend": backend, "n_plus_1_doubles": n + 1, "reps": reps}

    def spread(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # (a) torch.distributed
    if world > 1:
        buf = torch.ones(n + 1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        small = torch.ones(100, dtype=torch.float64, device=buf.device)
        gath = torch.empty(100 * world, dtype=torch.float64, device=buf.device)
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        out["torch_allreduce_us"] = spread(1e6 * (time.perf_counter() - t0) / reps)
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_gather_into_tensor(gath, small)
        torch.cuda.synchronize()
        out["torch_allgather_100_us"] = spread(1e6 * (time.perf_counter() - t0) / reps)
    # (b) the library's communicator: the mih_comm callbacks (device buffer: all-reduce; host buffer: all-gather)
    try:
        comm = D.NativeComm(0, 1, device=local)
        seen, path = comm.info()
        out["rccl_ranks_seen"], out["librccl"] = seen, path
        cstruct = C.cast(comm.pointer(), C.POINTER(api._Comm)).contents
        allreduce = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)(cstruct.allreduce)    # mih_comm::allreduce
        allgather = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)(cstruct.allgather)              # mih_comm::allgather
        dbuf = torch.ones(n + 1, dtype=torch.float64, device=dev)
        for _ in range(5):
            assert allreduce(cstruct.user, C.c_void_p(dbuf.data_ptr()), n + 1, 0, 1) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            assert allreduce(cstruct.user, C.c_void_p(dbuf.data_ptr()), n + 1, 0, 1) == 0
        out["library_allreduce_us"] = spread(1e6 * (time.perf_counter() - t0) / reps)
        send = np.ones(100)
        recv = np.zeros(100 * world)
        t0 = time.perf_counter()
        for _ in range(reps):
            assert allgather(cstruct.user, send.ctypes.data_as(C.c_void_p), 100, recv.ctypes.data_as(C.c_void_p)) == 0
        out["library_allgather_100_us"] = spread(1e6 * (time.perf_counter() - t0) / reps)
        out["library_sum_check"] = float(dbuf[0].item())          # world ** (reps + 5) would overflow: informational only
        comm.close()
    except Exception as e:      # noqa: BLE001
        out["library_error"] = repr(e)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
