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


def test_eight_rank_gather_of_fold_losses(tmp_path):
    """The shape of the driver's 8-GPU run (BASELINE configs[3]: 5 folds x 20 sparsity levels over eight ranks): every (fold, k)
    combination is owned by exactly one rank, 13 or 12 each, and every rank ends up with the whole loss matrix."""
    from mendeliht_amd import dist as D
    q, npath, world = 5, 20, 8
    shares = [list(D.shard_combinations(q, npath, r, world)) for r in range(world)]
    assert sorted(len(s_) for s_ in shares) == [12, 12, 12, 12, 13, 13, 13, 13]
    assert sorted(i for s_ in shares for i in s_) == list(range(q * npath))
    mp.spawn(_worker, args=(world, _free_port(), q, npath, str(tmp_path)), nprocs=world, join=True)
    full = np.arange(1.0, q * npath + 1).reshape(q, npath)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"r{r}.npy"), full)


def _comm_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import ctypes as C

    import torch.distributed as dist
    from mendeliht_amd import dist as D

    D.init_from_env(backend="gloo")
    lo, cnt = D.column_block(1000, rank, world)
    comm = D.ColumnComm(lo, 1000)
    assert (comm.rank, comm.world) == (rank, world) and comm.pointer() != 0
    # the callbacks exactly as the library invokes them (host buffers; no GPU in this test)
    v = np.arange(6, dtype=np.float64) * (rank + 1)
    assert comm._ar(None, v.ctypes.data_as(C.c_void_p), 6, 0, 0) == 0                     # sum
    mx = np.array([float(rank), -float(rank)])
    assert comm._ar(None, mx.ctypes.data_as(C.c_void_p), 2, 1, 0) == 0                    # max
    send = np.array([10.0 * rank + 1, 10.0 * rank + 2])
    recv = np.zeros(2 * world)
    assert comm._ag(None, send.ctypes.data_as(C.c_void_p), 2, recv.ctypes.data_as(C.c_void_p)) == 0
    np.save(os.path.join(out_dir, f"c{rank}.npy"), np.concatenate([v, mx, recv, [lo, cnt]]))
    dist.barrier()
    dist.destroy_process_group()


def test_column_comm_callbacks_two_ranks(tmp_path):
    world = 2
    mp.spawn(_comm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"c{r}.npy")
        assert np.array_equal(got[:6], np.arange(6.0) * 3)           # (1 + 2) x arange
        assert np.array_equal(got[6:8], [1.0, 0.0])
        assert np.array_equal(got[8:12], [1.0, 2.0, 11.0, 12.0])
    blocks = [np.load(tmp_path / f"c{r}.npy")[12:] for r in range(world)]
    assert blocks[0][0] == 0 and blocks[0][0] + blocks[0][1] == blocks[1][0] and blocks[1][0] + blocks[1][1] == 1000
    assert blocks[0][1] % 32 == 0


def test_column_block_covers_every_column_and_leaves_no_rank_empty():
    """The column sharding of a single fit: contiguous blocks that tile 0..p, on 32-column boundaries whenever there are at least
    as many column groups as ranks, and never an empty block while p >= world (found by the seeded random cases of
    tests/sharded_worker.py: p = 40 on 3 ranks used to give the last rank no column, which mih_snp_create refuses)."""
    import pytest

    from mendeliht_amd import dist as D
    from mendeliht_amd.api import MendelIHTError
    for p in (1, 2, 3, 5, 31, 32, 33, 40, 64, 65, 100, 127, 128, 129, 1000, 2300, 10_000, 1_000_003):
        for w in (1, 2, 3, 4, 8):
            if p < w:
                with pytest.raises(MendelIHTError):
                    D.column_block(p, 0, w)
                continue
            blocks = [D.column_block(p, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][0] + blocks[-1][1] == p, (p, w, blocks)
            assert all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(w - 1)), (p, w, blocks)
            assert all(c > 0 for _, c in blocks), (p, w, blocks)
            if (p + 31) // 32 >= w:
                assert all(lo % 32 == 0 for lo, _ in blocks), (p, w, blocks)
                sizes = [c for _, c in blocks]
                assert max(sizes) - min(sizes) <= 32 + 31, (p, w, blocks)


def _allgather_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import ctypes as C

    import torch.distributed as dist
    import mendeliht_amd as m
    from mendeliht_amd import api
    from mendeliht_amd import dist as D

    D.init_from_env(backend="gloo")
    q, npath = 5, 20
    full = np.arange(1.0, q * npath + 1).reshape(q, npath) * 1.000000123
    mine = np.zeros_like(full)
    for i in D.shard_combinations(q, npath, rank, world):
        mine.flat[i] = full.flat[i]
    # mih_cv_allgather (the library's gather of the cross-validation losses) through a communicator whose two exchanges are the
    # torch.distributed callbacks: the entry point itself makes no HIP call, so it runs here without a GPU
    comm = D.ColumnComm(0, 1)
    got = D.gather_losses_native(comm)(mine)
    # ordered sums: rank 0 + rank 1 + ... on EVERY rank, whatever the backend's all-reduce does
    oc = D.ColumnComm(0, 1, ordered_sum=True)
    v = np.array([0.1, 1e16, -1e16, 3.0]) * (rank + 1) + np.array([rank * 1e-3, 1.0, 2.0, -rank])
    w = v.copy()
    assert oc._ar(None, w.ctypes.data_as(C.c_void_p), w.size, 0, 0) == 0
    assert api.lib().mih_cv_allgather(None, None, 0) != 0                 # null communicator: refused, not crashed
    np.save(os.path.join(out_dir, f"g{rank}.npy"), got)
    np.save(os.path.join(out_dir, f"o{rank}.npy"), np.concatenate([w, v]))
    dist.barrier()
    dist.destroy_process_group()


def test_library_gather_of_cv_losses_and_ordered_sums_on_cpu(tmp_path):
    """mih_cv_allgather at world 2 and 3 over gloo (the N > 1 path of a cross-validation whose one exchange runs inside the
    library), and ColumnComm(ordered_sum=True): the sum every rank ends up with is rank 0 + rank 1 + rank 2 in that order."""
    for world in (2, 3):
        d = tmp_path / f"w{world}"
        d.mkdir()
        mp.spawn(_allgather_worker, args=(world, _free_port(), str(d)), nprocs=world, join=True)
        full = np.arange(1.0, 101.0).reshape(5, 20) * 1.000000123
        parts = [np.load(d / f"o{r}.npy") for r in range(world)]
        want = parts[0][4:].copy()
        for r in range(1, world):
            want = want + parts[r][4:]                                    # rank order
        for r in range(world):
            assert np.array_equal(np.load(d / f"g{r}.npy"), full)
            assert np.array_equal(parts[r][:4], want), (world, r)
