"""Residuals per fused pass beyond 12: the 16x16x128 ring kernel with 5 (15 residuals) and 6 (18 residuals) B operands
(MENDELIHT_XTV_MAX_OPS), ms per pass and per residual at n=500k, p=1M, plus bit-equality of every residual with the
12-per-pass split."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
for rnd in range(2):
    for mm in (10, 12, 13, 15, 16, 18, 24, 30, 36):
        ms, cs = x.bench_xtv_batched(mm, max_fused=4, iters=3, warmup=1)
        print(f"round {rnd} max_ops={os.environ.get('MENDELIHT_XTV_MAX_OPS', '5')} m={mm:2d}: {ms:7.2f} ms  {ms / mm:5.2f} ms/residual  checksum {cs:.12e}", flush=True)
