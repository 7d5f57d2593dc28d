"""Timing probe: the 4-operand X'r pass with the B operand read as FP6 (multi-variant 9; the output is NOT X'r)
against the FP4 default, to see whether a denser FP6 digit set could pay on the power-bound pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
for mv in (0, 9, 0, 9):
    m.lib().mih_set_xtv_multi_variant(mv)
    ms, cs = x.bench_xtv_batched(8, max_fused=4, iters=5, warmup=2)
    print(f"multi-variant {mv}: 4 operands {ms:8.2f} ms", flush=True)
m.lib().mih_set_xtv_multi_variant(0)
