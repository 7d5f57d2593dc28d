import sys, time, numpy as np, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from oracle import oracle as O
from conftest import make_bed
rng=np.random.default_rng(1)
n,p=200000,16384
cols=make_bed(rng,n,1024)
cols=np.tile(cols,(16,1))
ox=O.Mat.from_bed_columns(cols,n)
r=rng.standard_normal(n)
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
for th in (1,8,32,64,128,256):
    O.set_threads(th)
    ox.xtv(r); t=time.time()
    for _ in range(3): ox.xtv(r)
    dt=(time.time()-t)/3
    print(th, f"{dt*1e3:.1f} ms  {cols.nbytes/dt/1e9:.2f} GB/s", flush=True)
