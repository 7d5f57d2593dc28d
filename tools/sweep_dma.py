"""The LDS-DMA ring kernels (k_xtv_dma, mih_probe_set_xtv_multi_variant 20.. and the FP6 defaults) against the register-staged
LDS kernels: bit-equality on small ragged matrices in every residual format, then timings at n=500k, p=1M.
usage: sweep_dma.py [check|time|single|all]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MENDELIHT_HIP_PROBES", "1")     # kernel-shape knobs / A-B switches: the measurement build
import mendeliht_amd as m

what = sys.argv[1] if len(sys.argv) > 1 else "all"
L = m.lib()
FP6_SHAPES = [0, 20, 22, 40, 41, 42, 43]
FP4_SHAPES = {428: [0, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 34, 35], 1316: [20, 22]}


def check():
    rng = np.random.default_rng(7)
    bad = 0
    for (n, p, miss) in ((1000, 700, 0.0), (5003, 3001, 0.02), (40_000, 2100, 0.0), (1153, 96, 0.1), (130, 33, 0.0)):
        x = m.SnpLinAlg.synthetic(n, p, seed=11, missing_rate=miss)
        for digits in (0, 4908, 428, 1316):
            m.set_xtv_digits(digits)
            fp6 = digits in (0, 4908)
            ref_variant = 6 if fp6 else 9                 # the register-staged default shapes
            shapes = FP6_SHAPES if fp6 else FP4_SHAPES[digits]
            for mm in ((18, 16, 13, 12, 10, 9, 7, 6, 4, 3, 1) if digits != 428 else (1, 2, 4)):
                R = np.asfortranarray(rng.standard_normal((n, mm)) * np.exp(rng.uniform(-20, 20, mm)))
                m.probe_set(multi_variant=ref_variant)
                ref = x.xtv(R)
                for mv in shapes:
                    m.probe_set(multi_variant=mv)
                    out = x.xtv(R)
                    same = np.array_equal(out, ref)
                    bad += not same
                    if not same:
                        d = np.abs(out - ref).max() / np.abs(ref).max()
                        print(f"n={n} p={p} digits={digits} m={mm} variant {mv}: MISMATCH max rel {d:.3e}", flush=True)
        print(f"checked n={n} p={p} miss={miss}", flush=True)
    m.set_xtv_digits(0)
    m.probe_set(multi_variant=0)
    print("bit-equality:", "OK" if bad == 0 else f"{bad} MISMATCHES", flush=True)
    return bad


def time_(x):
    for rnd in range(2):                  # interleaved rounds in one process
        for mv in [6] + FP6_SHAPES:
            m.probe_set(multi_variant=mv)
            for mm in (12, 9, 6, 3):
                ms, cs = x.bench_xtv_batched(mm, max_fused=4, iters=4, warmup=1)
                B = x.algorithmic_bytes(mm)
                print(f"round {rnd} variant {mv:2d} m={mm:2d}: {ms:7.2f} ms  {B / ms / 1e6:6.0f} GB/s ({B / ms / 8e9 * 100:5.1f} % of 8 TB/s)  checksum {cs:.12e}", flush=True)
    m.probe_set(multi_variant=0)


def single(x):
    """the single-fit pass (format 428, one FP4 operand): LDS-DMA shapes against the register-staged default"""
    B = x.algorithmic_bytes(1)
    for rnd in range(3):
        for mv in [9] + FP4_SHAPES[428]:
            m.probe_set(multi_variant=mv)
            ms, cs = x.bench_xtv_batched(1, max_fused=4, iters=6, warmup=1)
            print(f"round {rnd} single-fit pass, variant {mv:2d}: {ms:7.3f} ms  {B / ms / 1e6:6.0f} GB/s ({B / ms / 8e9 * 100:5.1f} % of 8 TB/s)  checksum {cs:.12e}", flush=True)
    m.probe_set(multi_variant=0)


if what in ("check", "all"):
    if check():
        sys.exit(1)
if what in ("time", "single", "all"):
    x = m.SnpLinAlg.synthetic(500_000, 1_000_000, seed=2024)
    if what in ("time", "all"):
        time_(x)
    if what in ("single", "all"):
        single(x)
