"""Multi-GPU sharding of the cross-validation loop (src/cross_validation.jl:98-121).

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on
CPU-only test boxes).  Every rank holds a full 2-bit replica of X in its own HBM and
evaluates the (fold, k) combinations whose fold-major index is congruent to its rank;
the only data-path exchange is ONE all-gather of the q x len(path) held-out losses.
"""
import os

import numpy as np


def shard_combinations(q, npath, rank, world):
    """Indices (fold-major, cross_validation.jl:217-223) of the combinations a rank owns."""
    return [i for i in range(q * npath) if i % world == rank]


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


def cv_iht_distributed(y, x, z=None, **kw):
    """cv_iht with the (fold, k) loop sharded over the ranks of the current process group."""
    import torch.distributed as dist

    from .api import cv_iht

    if dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    return cv_iht(y, x, z, rank=rank, world=world, reduce=gather_losses, **kw)
