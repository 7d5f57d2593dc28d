"""Time fused multi-operand X'r passes for every kernel shape (mih_probe_set_xtv_multi_variant).
usage: sweep_multi.py [nops] [variants...]   (nops = 2 or 4 B operands per pass)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
nops = int(sys.argv[1]) if len(sys.argv) > 1 else 4
variants = [int(a) for a in sys.argv[2:]] or (list(range(0, 7)) if nops == 4 else list(range(0, 6)))
for mv in variants:
    m.probe_set(multi_variant=mv)
    ms, cs = x.bench_xtv_batched(nops, max_fused=nops, iters=3, warmup=1)
    B = x.algorithmic_bytes(nops)
    print(f"{nops} operands, multi-variant {mv:2d}: {ms:8.2f} ms  ({ms / nops:6.2f} ms/RHS)  {B / ms / 1e6:7.0f} GB/s algorithmic  checksum {cs:.12e}", flush=True)
m.probe_set(multi_variant=0)
