"""A/B of two builds of the library in alternating processes on one box: passes of m residuals at n = 500k, p = 1M.
usage: ab_libs.py OLD.so NEW.so [rounds]"""
import os, subprocess, sys
old, new = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
snippet = r'''
import os, sys
sys.path.insert(0, %r)
import mendeliht_amd as m
x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
out = []
for mm in (1, 12, 13, 18):
    ms, cs = x.bench_xtv_batched(mm, iters=6, warmup=2)
    out.append(f"m={mm}: {ms:6.2f} ms ({cs:.9e})")
print("  ".join(out), flush=True)
''' % ROOT
for rnd in range(rounds):
    for name, lib in (("old", old), ("new", new)):
        env = dict(os.environ, MENDELIHT_HIP_LIB=lib, MENDELIHT_HIP_PROBES="1")
        r = subprocess.run([sys.executable, "-c", snippet], env=env, capture_output=True, text=True)
        print(f"round {rnd} {name}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
