"""Run the X'r pass in a loop for a few seconds (for sampling clocks/power with rocm-smi alongside)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
nrhs, mv, secs = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
digits = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # residual format (include/mendeliht_hip.h), 0 = the default of a fused pass
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
m.probe_set(multi_variant=mv)
print("ready", flush=True)
t0 = time.time()
while time.time() - t0 < secs:
    ms, _ = x.bench_xtv_batched(nrhs, max_fused=4, iters=100, warmup=0, xtv_digits=digits) if nrhs > 1 or digits else x.bench_xtv(iters=100, warmup=0)
    print(f"nrhs={nrhs} {ms:.2f} ms/pass", flush=True)
