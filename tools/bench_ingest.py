"""Host .bed columns -> device matrix (mih_snp_create: upload + transcode + column statistics): GB/s."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
n, p = 500_000, int(os.environ.get("MIH_P", 65_536))
for miss in (0.0, 0.01):
    xs = m.SnpLinAlg.synthetic(n, p, seed=5, missing_rate=miss)
    cols = xs.export_bed()
    del xs
    for rep in range(2):
        t0 = time.perf_counter()
        x = m.SnpLinAlg(cols, n=n, center=True, scale=True, impute=True)
        dt = time.perf_counter() - t0
        print(f"missing={miss}: {cols.nbytes / 1e9:.2f} GB in {dt:.2f} s = {cols.nbytes / dt / 1e9:.1f} GB/s", flush=True)
        del x
