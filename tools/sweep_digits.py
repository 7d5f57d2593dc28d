"""Multi-RHS X'r passes in every residual format (the xtv_digits of a call): 0 = 10 base-49 FP6 digits, three residuals per B
operand (default); 1316 = 16 base-13 FP4 digits, two per operand; 428 = 28 base-4 digits, one per operand;
4908 / 1308 = the four-per-operand fast formats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
for digits in (0, 1316, 428, 4908, 1308):
    m.set_xtv_digits(digits)
    for nrhs in (1, 2, 3, 4, 6, 8, 10, 12, 13, 16):
        ms, cs = x.bench_xtv_batched(nrhs, max_fused=4, iters=3, warmup=1)
        print(f"digits={digits} m={nrhs:2d}: {ms:8.2f} ms  ({ms / nrhs:6.2f} ms/RHS)  checksum {cs:.9e}", flush=True)
m.set_xtv_digits(0)
