"""Does the 4-RHS X'r kernel time depend on the operand data (zero vs random residuals)?  If so the
matrix pipe is power/clock limited rather than issue limited."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mendeliht_amd as m
n, p = 500_000, 1_000_000
x = m.SnpLinAlg.synthetic(n, p, seed=2024)
rng = np.random.default_rng(1)
Rz = np.zeros((n, 4), order="F")
Rr = np.asfortranarray(rng.standard_normal((n, 4)))
Rs = np.asfortranarray(np.round(rng.standard_normal((n, 4))))      # few non-zero digits
for mv in [int(a) for a in sys.argv[1:]] or [0, 8, 7]:
    m.lib().mih_set_xtv_multi_variant(mv)
    for name, R in (("zeros", Rz), ("small-int", Rs), ("random", Rr)):
        x.xtv(R)
        m.profile_read(reset=True); m.profile_enable(True)
        for _ in range(3):
            x.xtv(R)
        m.profile_enable(False)
        ms, k = m.profile_read(reset=True)
        print(f"multi-variant {mv}: r={name:9s} {ms / k:7.2f} ms per 4-RHS pass ({k} launches)", flush=True)
