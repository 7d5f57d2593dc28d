"""Single-operand X'r pass: FP6 digit planes (default residual format) in several launch shapes against the FP4
base-13 format, alternating so that clock drift cancels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
for rep in range(2):
    for digits, mv in ((1316, 0), (0, 0), (0, 10), (0, 11), (0, 12), (0, 13), (0, 14), (0, 15), (428, 0)):
        m.set_xtv_digits(digits)
        m.probe_set(multi_variant=mv)
        ms, cs = x.bench_xtv_batched(1, max_fused=4, iters=10, warmup=2)
        print(f"format {digits:4d} shape {mv:2d}: {ms:8.3f} ms", flush=True)
m.probe_set(multi_variant=0)
m.set_xtv_digits(0)
