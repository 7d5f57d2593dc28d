"""Single-operand X'r pass: the round-1 register-staged LDS shapes (mih_probe_set_xtv_multi_variant 9 = round 1's default
<1,1,4>, 10..14; 0 = the round-2 library default, the LDS-DMA ring k_xtv_dma<1,2,4,8>; tools/sweep_dma.py sweeps the ring shapes)
against the best per-wave-load shape (mih_probe_set_xtv_variant 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
B = x.algorithmic_bytes(1)
for mv in (0, 9, 10, 11, 12, 13, 14, 15):        # 15 = load-only probe (no MFMAs): the memory ceiling of this access shape
    m.probe_set(multi_variant=mv)
    ms, cs = x.bench_xtv_batched(1, max_fused=4, iters=5, warmup=1)
    print(f"LDS shape {mv:2d}: {ms:7.3f} ms  {B / ms / 1e6:6.0f} GB/s  checksum {cs:.12e}", flush=True)
m.probe_set(multi_variant=0)
ms, cs = x.bench_xtv_batched(1, max_fused=4, iters=5, warmup=1, variant=2)
print(f"per-wave <4,4>: {ms:7.3f} ms  {B / ms / 1e6:6.0f} GB/s  checksum {cs:.12e}", flush=True)
