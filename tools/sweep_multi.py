"""Time the fused 4-RHS X'r pass for every kernel shape (mih_set_xtv_multi_variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
variants = [int(a) for a in sys.argv[1:]] or list(range(0, 7))
for mv in variants:
    m.lib().mih_set_xtv_multi_variant(mv)
    ms, cs = x.bench_xtv_batched(4, max_fused=4, iters=3, warmup=1)
    B = x.algorithmic_bytes(4)
    print(f"multi-variant {mv:2d}: {ms:8.2f} ms  ({ms / 4:6.2f} ms/RHS)  {B / ms / 1e6:7.0f} GB/s algorithmic  checksum {cs:.12e}", flush=True)
