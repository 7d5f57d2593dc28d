"""Round-1 launch shapes of the 4-operand FP6 X'r pass (mih_probe_set_xtv_multi_variant 1..6 = register-staged shapes, 6 =
round 1's default <4,2,1,8>; 0 = the round-2 library default k_xtv_dma16<4,2,8,4>, see tools/sweep_dma.py).  Round 1 read: 0 =
<4,2,1,8>, the default, 1 = <4,2,2,8>) and
the 3-operand passes of both digit formats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
for mv in (0, 6, 1, 2, 3, 4, 5, 0):
    m.probe_set(multi_variant=mv)
    ms, cs = x.bench_xtv_batched(12, max_fused=4, iters=4, warmup=1)
    print(f"FP6 4 operands (12 residuals), shape {mv}: {ms:8.2f} ms  checksum {cs:.9e}", flush=True)
m.probe_set(multi_variant=0)
for digits, mm in ((0, 9), (0, 6), (0, 3), (1316, 6), (1316, 4), (1316, 8)):
    m.set_xtv_digits(digits)
    ms, cs = x.bench_xtv_batched(mm, max_fused=4, iters=4, warmup=1)
    print(f"format {digits}: {mm} residuals {ms:8.2f} ms", flush=True)
m.set_xtv_digits(0)
