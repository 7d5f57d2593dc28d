import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MENDELIHT_HIP_PROBES"] = "1"
import mendeliht_amd as m
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
for rnd in range(4):
    for order in ((8, 4, 5), (4, 5, 8), (5, 8, 4)):
        row = []
        for s in order:
            os.environ["MENDELIHT_XTV_SLICES"] = str(s)
            ms, cs = x.bench_xtv(iters=10, warmup=2)
            row.append(f"S={s}: {ms:6.3f}")
        print(f"round {rnd} m=1  " + "  ".join(row), flush=True)
