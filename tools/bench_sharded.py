"""Column-sharded single fit: time per IHT iteration with the SNP columns split over the ranks.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 \
         --master-port 29533 tools/bench_sharded.py            # one GPU per rank, RCCL
  MIH_ONE_DEVICE=1 MIH_BACKEND=gloo ... (all ranks on GPU 0: functional check on a 1-GPU box)

Sizes via MIH_N / MIH_P / MIH_K (default n=500k, p=1M, k=200: BASELINE configs[2] split over W GPUs).
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
from mendeliht_amd import dist as D

n, p, k = int(os.environ.get("MIH_N", 500_000)), int(os.environ.get("MIH_P", 1_000_000)), int(os.environ.get("MIH_K", 200))
steps = int(os.environ.get("MIH_STEPS", 10))
rank, world, local = D.init_from_env(backend=os.environ.get("MIH_BACKEND"))
dev = 0 if os.environ.get("MIH_ONE_DEVICE") else local
import torch
torch.cuda.set_device(dev)
lo, cnt = D.column_block(p, rank, world)
x = m.SnpLinAlg.synthetic(n, cnt, seed=2024, device=dev, col_offset=lo)
rng = np.random.default_rng(2025)                       # same stream on every rank: replicated y
supp = np.sort(rng.choice(p, size=k, replace=False))
beta = rng.standard_normal(k)
mine = (supp >= lo) & (supp < lo + cnt)
comm = (D.NativeComm if os.environ.get("MIH_NATIVE") else D.ColumnComm)(lo, p, device=dev)     # MIH_NATIVE=1: the library's own RCCL communicator
xb = x.xv_sparse(supp[mine] - lo, beta[mine])
if world > 1:
    import torch.distributed as dist
    t = torch.from_numpy(xb)
    if dist.get_backend() == "nccl":
        t = t.cuda(); dist.all_reduce(t); xb = t.cpu().numpy()
    else:
        dist.all_reduce(t)
y = xb + 1.0 + rng.standard_normal(n)
sess = m.IHTSession(y, x, None, k=k, d=m.Normal(), l=m.IdentityLink(), comm=comm)
for _ in range(2):
    sess.step()
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
for _ in range(steps):
    logl, bt, tol = sess.step()
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
el = time.perf_counter() - t0
b, _ = sess.model()
found = int(np.intersect1d(np.flatnonzero(b) + lo, supp).size)
if world > 1:
    f = torch.tensor([float(found)])
    if dist.get_backend() == "nccl":
        f = f.cuda()
    dist.all_reduce(f); found = int(f.item())
if rank == 0:
    print(f"world={world} n={n} p={p} k={k}: {1e3 * el / steps:.2f} ms/iteration ({steps / el:.1f} it/s), "
          f"logl={logl!r}, true effects recovered {found}/{k}, block={cnt} columns/rank", flush=True)
sess.close()
if world > 1:
    dist.destroy_process_group()
