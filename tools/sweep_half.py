"""Fused passes whose last operand is half empty (m = 3 j + 1 residuals): time per pass with the second 16-column fragment
of that operand left out (default) or multiplied anyway (MENDELIHT_XTV_NO_HALF=1, read once per process).
usage: sweep_half.py            (run it twice, with and without the environment variable)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m

x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
tag = "whole operands" if os.environ.get("MENDELIHT_XTV_NO_HALF") else "half operands "
for rnd in range(2):
    for mm in (1, 4, 7, 10, 13, 12, 15):
        ms, cs = x.bench_xtv_batched(mm, max_fused=4, iters=4, warmup=1)
        print(f"{tag} round {rnd} m={mm:2d}: {ms:7.2f} ms ({ms / mm:5.2f} ms per residual)  checksum {cs:.12e}", flush=True)
