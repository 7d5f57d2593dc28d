"""The N>1 path on CPU: world_size-2 gloo process group exercising the (fold,k) sharding and the
single all-gather of held-out losses that cv_iht_distributed performs over RCCL on the GPU box."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q, npath, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import mendeliht_amd  # noqa: F401
    from mendeliht_amd import dist as D

    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    full = np.arange(1.0, q * npath + 1).reshape(q, npath)          # stand-in for the per-combination losses
    mine = np.zeros_like(full)
    for i in D.shard_combinations(q, npath, rank, world):
        mine.flat[i] = full.flat[i]                                  # what mih_cv_iht fills on this rank
    total = D.gather_losses(mine)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_of_fold_losses(tmp_path):
    q, npath, world = 5, 20, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, q, npath, str(tmp_path)), nprocs=world, join=True)
    full = np.arange(1.0, q * npath + 1).reshape(q, npath)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"r{r}.npy"), full)
