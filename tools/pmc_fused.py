"""A few fused 12-residual X'R passes (n=500k, p=1M) with one kernel shape, for rocprofv3 --pmc / --kernel-trace.
usage: pmc_fused.py MULTI_VARIANT [residuals=12] [passes=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m
mv = int(sys.argv[1]); mm = int(sys.argv[2]) if len(sys.argv) > 2 else 12; it = int(sys.argv[3]) if len(sys.argv) > 3 else 3
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
m.probe_set(multi_variant=mv)
ms, cs = x.bench_xtv_batched(mm, max_fused=4, iters=it, warmup=1)
print(f"variant {mv} m={mm}: {ms:.2f} ms/pass checksum {cs:.12e}", flush=True)
