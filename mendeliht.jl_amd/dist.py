"""Multi-GPU paths: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on
CPU-only test boxes).

1. Cross-validation (src/cross_validation.jl:98-121): every rank holds a full 2-bit replica of X in its
   own HBM and evaluates the (fold, k) combinations whose fold-major index is congruent to its rank;
   the only data-path exchange is ONE all-gather of the q x len(path) held-out losses.
2. Column-sharded single fit (`ColumnComm`, `fit_iht_sharded`): every rank holds a contiguous block of
   the SNP columns; per iteration the library asks for two n-vector sums (X_S b_S of update_xb!, X_S g_S
   of iht_stepsize!), one small all-gather (the top-k candidates of project_k!) and a few scalars.  The
   X'r pass itself needs no exchange.
"""
import ctypes as C
import os

import numpy as np


def shard_combinations(q, npath, rank, world, path=None):
    """Indices (fold-major, cross_validation.jl:217-223) of the combinations a rank owns under the library's sharding rule
    (mih_cv_assignment: round-robin over the combinations sorted by model size, so every rank gets every size class).
    `path` defaults to 1:npath."""
    from .api import cv_assignment
    rank_of = cv_assignment(range(1, npath + 1) if path is None else path, q, world).ravel()
    return [i for i in range(q * npath) if rank_of[i] == rank]


def init_from_env(backend=None):
    """Join the process group torchrun set up (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_losses(raw):
    """One all-gather of each rank's (mostly zero) loss matrix; returns their sum."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return raw
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    raw = np.ascontiguousarray(raw, dtype=np.float64)
    mine = torch.from_numpy(raw.ravel().copy()).to(dev)
    out = torch.empty(dist.get_world_size() * mine.numel(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, mine)
    return out.view(dist.get_world_size(), -1).sum(dim=0).cpu().numpy().reshape(raw.shape)


def gather_losses_native(comm):
    """reduce= for cv_iht over the library's own communicator (mih_cv_allgather: one ncclAllGather, no torch collective)."""
    def reduce(raw):
        from .api import _check, _p, lib
        out = np.ascontiguousarray(raw, dtype=np.float64).copy()
        _check(lib().mih_cv_allgather(comm.pointer(), _p(out), out.size))
        return out.reshape(np.shape(raw))
    return reduce


def cv_iht_distributed(y, x, z=None, native=False, **kw):
    """cv_iht with the (fold, k) loop sharded over the ranks of the current process group.  native=True: the one exchange
    runs inside the library over its own RCCL communicator (mih_cv_allgather) -- what a Julia multi-process run uses."""
    import torch.distributed as dist

    from .api import cv_iht

    if dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    if not native:
        return cv_iht(y, x, z, rank=rank, world=world, reduce=gather_losses, **kw)
    comm = NativeComm(0, 1, device=getattr(x, "device", 0))
    try:
        return cv_iht(y, x, z, rank=rank, world=world, reduce=gather_losses_native(comm), **kw)
    finally:
        comm.close()


class ColumnComm:
    """The `mih_comm` of one rank of a column-sharded fit: ctypes callbacks over torch.distributed.

    Device buffers handed over by the library are wrapped in place (`__cuda_array_interface__`) and
    reduced by RCCL; with the gloo backend they are staged through host memory.
    """

    def __init__(self, col_offset, p_global, device=0, group=None, ordered_sum=False):
        """ordered_sum: sums are taken in RANK ORDER on every rank (one all-gather + a local sum) instead of by the backend's
        all-reduce, whose ring sums every chunk in a different order -- results then do not depend on the backend or its
        algorithm (for more than two ranks a floating-point sum depends on the order), at world times the traffic."""
        import torch.distributed as dist

        from .api import _ALLGATHER, _ALLREDUCE, _Comm

        self.group = group
        self.device = device
        self.ordered_sum = bool(ordered_sum)
        self.error = None
        on = dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        self.backend = dist.get_backend(group) if on else None
        self._ar = _ALLREDUCE(self._allreduce)
        self._ag = _ALLGATHER(self._allgather)
        self._c = _Comm(self.rank, self.world, int(col_offset), int(p_global),
                        C.cast(self._ar, C.c_void_p), C.cast(self._ag, C.c_void_p), None)

    def pointer(self):
        return C.addressof(self._c)

    # -- callbacks (must not raise through ctypes: report failure by return code) -----------------
    def _allreduce(self, _user, buf, count, op, on_device):
        try:
            import torch
            import torch.distributed as dist

            if self.world == 1:
                return 0
            rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
            if self.ordered_sum and op == 0:
                return self._allreduce_ordered(buf, count, on_device)
            if on_device:
                t = torch.as_tensor(_DevArray(buf, count), device=f"cuda:{self.device}")
                if self.backend == "nccl":
                    dist.all_reduce(t, op=rop, group=self.group)
                else:
                    hcopy = t.cpu()
                    dist.all_reduce(hcopy, op=rop, group=self.group)
                    t.copy_(hcopy)
                torch.cuda.synchronize(self.device)
            else:
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(count,))
                t = torch.from_numpy(arr)
                if self.backend == "nccl":
                    d = t.to(f"cuda:{self.device}")
                    dist.all_reduce(d, op=rop, group=self.group)
                    arr[:] = d.cpu().numpy()
                else:
                    dist.all_reduce(t, op=rop, group=self.group)
            return 0
        except Exception as e:      # noqa: BLE001
            self.error = e
            return 1

    def _allreduce_ordered(self, buf, count, on_device):
        import torch
        import torch.distributed as dist

        dev = f"cuda:{self.device}" if self.backend == "nccl" else "cpu"
        if on_device:
            t = torch.as_tensor(_DevArray(buf, count), device=f"cuda:{self.device}")
            mine = t if self.backend == "nccl" else t.cpu()
        else:
            arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(count,))
            t = None
            mine = torch.from_numpy(arr.copy()).to(dev)
        out = torch.empty(count * self.world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(out, mine.contiguous(), group=self.group)
        parts = out.view(self.world, count)
        acc = parts[0].clone()
        for r in range(1, self.world):                      # rank 0 first, then 1, 2, ...: the same order on every rank
            acc += parts[r]
        if on_device:
            t.copy_(acc)
            torch.cuda.synchronize(self.device)
        else:
            arr[:] = acc.cpu().numpy()
        return 0

    def _allgather(self, _user, send, count, recv):
        try:
            import torch
            import torch.distributed as dist

            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_double)), shape=(count,))
            dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_double)), shape=(count * self.world,))
            if self.world == 1:
                dst[:] = src
                return 0
            dev = f"cuda:{self.device}" if self.backend == "nccl" else "cpu"
            mine = torch.from_numpy(src.copy()).to(dev)
            out = torch.empty(count * self.world, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(out, mine, group=self.group)
            dst[:] = out.cpu().numpy()
            return 0
        except Exception as e:      # noqa: BLE001
            self.error = e
            return 1


def _last_error_text():
    from .api import lib
    buf = C.create_string_buffer(512)
    lib().mih_last_error(buf, 512)
    return buf.value.decode(errors="replace")


class NativeComm:
    """The library's own RCCL communicator as the `mih_comm` of a column-sharded fit (mih_comm_create_rccl): the two
    exchanges run inside the library (ncclAllReduce / ncclAllGather over xGMI), no callback into Python.  One process per
    GPU; the 128-byte unique id is made by rank 0 and broadcast over the torch.distributed group (any backend)."""

    def __init__(self, col_offset, p_global, device=0, group=None):
        import torch
        import torch.distributed as dist

        from .api import _check, lib

        on = dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        self.error = None
        self._h = C.c_void_p(None)
        on_gpu = on and dist.get_backend(group) == "nccl"

        def everyone(ok):
            #  a rank-local failure must not leave the other ranks inside the next collective: agree first
            if self.world == 1:
                return bool(ok)
            f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=f"cuda:{device}" if on_gpu else "cpu")
            dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
            return bool(int(f.item()))

        # 1. every rank proves that it can load librccl and ask it for an id (a local call; only rank 0's id is used)
        uid = (C.c_char * 128)()
        rc = lib().mih_rccl_unique_id(uid)
        mine = None if rc == 0 else _last_error_text()
        if not everyone(rc == 0):
            raise RuntimeError("mih_rccl_unique_id failed on " + ("this rank: " + mine if mine else "another rank"))
        # 2. rank 0's id to everyone
        if self.world > 1:
            t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
            if on_gpu:
                t = t.to(f"cuda:{device}")
            dist.broadcast(t, src=0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        # 3. ncclCommInitRank is itself collective; a rank on which it RETURNS an error (rather than hanging) is reported to all
        rc = lib().mih_comm_create_rccl(uid, self.rank, self.world, int(device), int(col_offset), int(p_global), C.byref(self._h))
        mine = None if rc == 0 else _last_error_text()
        if not everyone(rc == 0):
            self.close()
            raise RuntimeError("mih_comm_create_rccl failed on " + ("this rank: " + mine if mine else "another rank"))

    def pointer(self):
        return self._h.value

    def info(self):
        """(ranks RCCL itself reports for this communicator, the librccl file that was loaded) -- mih_comm_info"""
        from .api import _check, lib
        seen, path = C.c_int32(-1), C.create_string_buffer(1024)
        _check(lib().mih_comm_info(self._h, C.byref(seen), path, 1024))
        return int(seen.value), path.value.decode(errors="replace")

    def close(self):
        if self._h:
            from .api import lib
            lib().mih_comm_destroy_rccl(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevArray:
    """A device pointer as a CUDA-array-interface object (zero-copy view for torch.as_tensor)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def column_block(p_global, rank, world):
    """(offset, size) of the contiguous block of SNP columns a rank owns: the 32-column groups (one MFMA column group) dealt out
    as evenly as they go, the earlier ranks taking the remainder; with fewer groups than ranks the columns themselves are dealt
    out, so that no rank is left without a column as long as p_global >= world (a shard of no columns is not a matrix:
    mih_snp_create refuses it)."""
    if p_global < world:
        from .api import ArgumentError
        raise ArgumentError(f"{world} ranks for {p_global} SNP columns: every rank needs at least one column")
    groups = (p_global + 31) // 32
    if groups >= world:
        base, rem = divmod(groups, world)
        lo = 32 * (rank * base + min(rank, rem))
        hi = 32 * ((rank + 1) * base + min(rank + 1, rem))
        return min(lo, p_global), min(hi, p_global) - min(lo, p_global)
    base, rem = divmod(p_global, world)
    lo = rank * base + min(rank, rem)
    return lo, base + (1 if rank < rem else 0)


def fit_iht_sharded(y, x_shard, z=None, *, col_offset, p_global, weight=None, native=False, ordered_sum=False, **kw):
    """fit_iht on a design matrix whose SNP columns are sharded over the ranks of the process group.

    x_shard holds columns [col_offset, col_offset + x_shard.p); y, z are replicated.  Returns the same
    IHTResult on every rank with the full-length beta assembled from the shards (a multivariate y -- r x n -- gives the
    mIHTResult with the r x p_global B).  native=True: the exchanges run
    inside the library over its own RCCL communicator (one GPU per rank) instead of through torch.distributed callbacks.
    """
    import torch
    import torch.distributed as dist

    from .api import fit_iht

    comm = (NativeComm(col_offset, p_global, device=x_shard.device) if native else
            ColumnComm(col_offset, p_global, device=x_shard.device, ordered_sum=ordered_sum))
    if weight is not None:
        weight = np.asarray(weight, dtype=np.float64)[col_offset:col_offset + x_shard.p]
    if kw.get("group") is not None:          # (round 6) group labels 1..G of ALL p_global columns: the shard takes its own; J and k stay global
        kw = dict(kw, group=np.asarray(kw["group"])[col_offset:col_offset + x_shard.p])
    try:
        res = fit_iht(y, x_shard, z, weight=weight, comm=comm, **kw)
    except Exception:
        if comm.error is not None:
            raise comm.error
        raise
    finally:
        #  the native communicator is torn down here, collectively and while every rank is still alive -- not by the
        # garbage collector at interpreter exit (ncclCommDestroy after the other ranks have gone can hang)
        if native:
            comm.close()
    mv = np.ndim(res.beta) == 2
    if mv:                          # multivariate (mIHTResult): beta is B, r x p, this shard's columns
        r = res.beta.shape[0]
        tr, col = np.nonzero(res.beta)
        mine = (tr, col + col_offset, res.beta[tr, col])
    else:
        nz = np.flatnonzero(res.beta)
        mine = (nz + col_offset, res.beta[nz])
    if comm.world > 1:
        parts = [None] * comm.world
        dist.all_gather_object(parts, mine)
    else:
        parts = [mine]
    if mv:
        B = np.zeros((r, p_global), order="F")
        for ti, gi, gv in parts:
            B[ti, gi] = gv
        res.beta = B
        return res
    beta = np.zeros(p_global)
    for gi, gv in parts:
        beta[gi] = gv
    res.beta = beta
    return res
