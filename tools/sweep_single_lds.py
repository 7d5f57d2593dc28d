"""Single-operand X'r pass with LDS-shared digit planes: kernel shapes x row-slice counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mendeliht_amd as m
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
for var, splits in ((2, 8), (11, 4), (12, 16)):          # k_xtv variant ids whose `splits` are 8 / 4 / 16
    for mv in (0, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19):
        m.lib().mih_set_xtv_multi_variant(mv)
        ms, cs = x.bench_xtv_batched(1, max_fused=4, iters=5, warmup=1, variant=var)
        print(f"splits={splits:2d} multi-variant {mv:2d}: {ms:7.3f} ms checksum {cs:.12e}", flush=True)
m.lib().mih_set_xtv_multi_variant(0)
