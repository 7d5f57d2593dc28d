"""Which half of the LDS-DMA fused pass sets its time?  Full kernel vs no-MFMA vs no-copy probes (results of the probes
are not X'R), 12 residuals at n=500k, p=1M, interleaved rounds in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
names = {20: "<4,4,4,4> full", 30: "<4,4,4,4> no MFMAs", 31: "<4,4,4,4> no copies", 22: "<4,2,8,4> full", 32: "<4,2,8,4> no MFMAs", 33: "<4,2,8,4> no copies", 0: "register-staged <4,2,1,8>"}
variants = [int(a) for a in sys.argv[1:]] or [0, 20, 30, 31, 22, 32, 33]
for rnd in range(2):
    for mv in variants:
        m.probe_set(multi_variant=mv)
        ms, cs = x.bench_xtv_batched(12, max_fused=4, iters=4, warmup=1)
        print(f"round {rnd} variant {mv:2d} {names.get(mv, ''):28s}: {ms:7.2f} ms", flush=True)
m.probe_set(multi_variant=0)
