"""Time every X'r kernel variant on one synthetic matrix (default: BASELINE configs[2] geometry)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=500_000)
ap.add_argument("--p", type=int, default=1_000_000)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--variants", type=str, default="")
a = ap.parse_args()
x = m.SnpLinAlg.synthetic(a.n, a.p, seed=2024)
B = x.algorithmic_bytes()
ids = [int(v) for v in a.variants.split(",")] if a.variants else list(range(64)) + list(range(100, 110))
for v in ids:
    try:
        m.probe_set(variant=v)
    except m.MendelIHTError:
        continue
    ms, cs = x.bench_xtv(v, iters=a.iters, warmup=1)
    print(f"variant {v:3d}: {ms:8.3f} ms  {B / ms / 1e6:8.1f} GB/s  ({B / ms / 1e6 / 80:5.1f}% of 8 TB/s)  checksum {cs:.9e}", flush=True)
