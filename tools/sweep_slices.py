"""Row slices of the fused X'R passes (MENDELIHT_XTV_SLICES, measurement build): every (column group, slice) work item pays a prologue
and an epilogue that grows with the operand count, so wide passes may prefer fewer slices than the single-fit pass's eight.
Alternated in one process at n = 500k, p = 1M."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MENDELIHT_HIP_PROBES"] = "1"
import mendeliht_amd as m
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
for rnd in range(3):
    for mm in (1, 6, 12, 13, 15, 18):
        row = []
        for s in (8, 4, 6, 3, 2):
            os.environ["MENDELIHT_XTV_SLICES"] = str(s)
            ms, cs = x.bench_xtv_batched(mm, iters=4, warmup=1)
            row.append(f"S={s}: {ms:6.2f}")
        print(f"round {rnd} m={mm:2d}  " + "  ".join(row), flush=True)
