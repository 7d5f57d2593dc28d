"""Same-box A/B of round 3's bank-conflict fix in k_xtv_dma16: the product kernel (single ds_read_b64 per fragment half) against
the same kernel with round 2's plain 8-byte loads, which the compiler pairs into ds_read2_b64 / ds_read2st64_b64 (measurement
build, multi-variant 41), interleaved in one process at n = 500k, p = 1M; 12 and 15 residuals; checksums must agree."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")
import mendeliht_amd as m
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
for rnd in range(3):
    for mm in (12, 15):
        row = []
        for mv, name in ((0, "single ds_read_b64 (round 3)"), (41, "paired ds_read2_b64 (round 2)")):
            m.probe_set(multi_variant=mv)
            ms, cs = x.bench_xtv_batched(mm, iters=4, warmup=1)
            row.append((name, ms, cs))
        assert row[0][2] == row[1][2], row
        print(f"round {rnd} m={mm}: " + "; ".join(f"{nm}: {ms:6.2f} ms" for nm, ms, _ in row) + f"  (checksum {row[0][2]:.9e}, identical)", flush=True)
m.probe_set(multi_variant=0)
