"""Does the 4-RHS X'r kernel time depend on the operand data (zero vs random residuals)?  If so the
matrix pipe is power/clock limited rather than issue limited."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(1)
NC = int(os.environ.get("PROBE_COLS", "12"))     # 12 residuals = 4 FP6 operands (default format)
Rz = np.zeros((n, NC), order="F")
Rr = np.asfortranarray(rng.standard_normal((n, NC)))
Rs = np.asfortranarray(np.round(rng.standard_normal((n, NC))))      # few non-zero digits
for mv in [int(a) for a in sys.argv[1:]] or [0, 8, 7]:
    m.probe_set(multi_variant=mv)
    for name, R in (("zeros", Rz), ("small-int", Rs), ("random", Rr)):
        x.xtv(R)
        m.profile_read(x, reset=True); m.profile_enable(x, True)
        for _ in range(3):
            x.xtv(R)
        m.profile_enable(x, False)
        ms, k = m.profile_read(x, reset=True)
        print(f"multi-variant {mv}: r={name:9s} {ms / k:7.2f} ms per pass ({k} launches)", flush=True)
