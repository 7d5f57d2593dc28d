"""BASELINE configs[3] end to end on W GPUs: cv_iht (Bernoulli/Logit, path=1:20, 5 folds) with the (fold,k)
fits sharded over the ranks of a torch.distributed job and ONE all-gather of the losses.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 \
         --master-port 29544 tools/bench_cv_dist.py          # one GPU per rank (RCCL)
  MIH_ONE_DEVICE=1 MIH_BACKEND=gloo ...                      # all ranks on GPU 0 (functional check)
Sizes via MIH_N / MIH_P (default 500k x 1M: every rank synthesises its own 125 GB replica of X)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mendeliht_amd as m
from mendeliht_amd import dist as D
hash_folds = m.hash_folds

n, p = int(os.environ.get("MIH_N", 500_000)), int(os.environ.get("MIH_P", 1_000_000))
rank, world, local = D.init_from_env(backend=os.environ.get("MIH_BACKEND"))
dev = 0 if os.environ.get("MIH_ONE_DEVICE") else local
import torch
torch.cuda.set_device(dev)
x = m.SnpLinAlg.synthetic(n, p, seed=2024, device=dev)          # same seed: identical replicas
rng = np.random.default_rng(2025)
supp = np.sort(rng.choice(p, 10, replace=False))
eta = x.xv_sparse(supp, rng.standard_normal(10) * 0.5)
y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
folds = hash_folds(n, 5)
if world > 1:
    import torch.distributed as dist
    dist.barrier()
t0 = time.perf_counter()
mse = D.cv_iht_distributed(y, x, None, path=range(1, 21), q=5, folds=folds, verbose=False, d=m.Bernoulli(), l=m.LogitLink())
dt = time.perf_counter() - t0
if rank == 0:
    print(f"cv_iht Bernoulli/Logit n={n} p={p} path=1:20 q=5 on {world} rank(s): {dt:.2f} s, best k = {int(np.argmin(mse)) + 1}", flush=True)
if world > 1:
    dist.destroy_process_group()
