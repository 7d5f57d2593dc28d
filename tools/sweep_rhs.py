"""Time m right-hand sides through the X'r pass with different fusion widths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
for nrhs in (1, 2, 4, 10):
    for fused in (1, 2, 4):
        if fused > nrhs: continue
        ms, cs = x.bench_xtv_batched(nrhs, max_fused=fused, iters=3, warmup=1)
        B = x.algorithmic_bytes(nrhs)
        print(f"m={nrhs:2d} fused<={fused}: {ms:8.2f} ms  ({ms / nrhs:6.2f} ms/RHS)  {B / ms / 1e6:7.0f} GB/s algorithmic  checksum {cs:.9e}", flush=True)
