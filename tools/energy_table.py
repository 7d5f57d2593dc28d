#!/usr/bin/env python3
"""tools/energy_table.py OUT.json -- joules per X'r pass (the measurement the "bound by the matrix pipe under the
package power cap" claim of DESIGN 3.1b rests on).  For each workload a child process loops it for ~7 s while this process samples
`rocm-smi --showpower --showclocks` (~3 samples/s; the first 2 s dropped: the clock settles): mean package power x time per pass =
J per pass, and J per MFMA against tools/mfma_rate doing the same instruction with its operands in registers (no memory traffic).
This process makes no HIP call."""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_file = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "energy.json")


def sample():
    try:
        t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        return None, None
    p = re.search(r"Power \(W\): ([0-9.]+)", t)
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
    return (float(p.group(1)) if p else None), (int(c.group(1)) if c else None)


def watch(cmd, ready=None, env=None, settle=2.0):
    log = open("/tmp/energy_child.log", "w")
    pr = subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, env=env)
    t0 = time.time()
    if ready:
        while pr.poll() is None and ready not in open("/tmp/energy_child.log").read():
            time.sleep(0.2)
    t1 = time.time()
    pw, ck = [], []
    while pr.poll() is None:
        p, c = sample()
        if p is not None and time.time() - t1 > settle:
            pw.append(p); ck.append(c or 0)
    pr.wait()
    return pw, ck, open("/tmp/energy_child.log").read()


def active_mean(pw):
    """package power while the kernel runs: the samples within 10 % of the largest (a sample that falls between two batches of
    launches of the looping child reads several hundred watts less)"""
    top = max(pw)
    act = [v for v in pw if v >= 0.9 * top]
    return sum(act) / len(act)


idle = [sample()[0] for _ in range(5)]
rows = []
env = dict(os.environ, MENDELIHT_HIP_PROBES="1")
# (residuals, digits, MFMA-equivalents of 32x32x64 per pass: operands x n_pad/64 row blocks x p/32 column groups)
n_pad, p = 500_096, 1_000_000
specs = [(19, 0), (16, 0), (13, 0), (10, 0), (6, 0), (3, 0), (24, 4908), (20, 4908), (1, 0)]
for m, dg in specs:
    pw, ck, txt = watch([sys.executable, os.path.join(ROOT, "tools", "spin_multi.py"), str(m), "0", "7"] + ([str(dg)] if dg else []), ready="ready", env=env)
    ms = [float(v) for v in re.findall(r"([0-9.]+) ms/pass", txt)]
    if not pw or not ms:
        rows.append({"residuals": m, "digits": dg, "error": txt[-300:]}); continue
    ms_pass = sorted(ms)[len(ms) // 2]
    per_op = 4 if dg == 4908 else 3
    cols = m * (8 if dg == 4908 else 10)
    ops = 1 if m == 1 and not dg else (cols + 31) // 32
    mfma = ops * (n_pad / 64) * (p / 32)
    W = active_mean(pw)
    rows.append({"workload": f"{'single-fit pass (428)' if m == 1 and not dg else 'fused pass'}, {m} residual(s), format {dg or ('428' if m == 1 else 4910)}",
                 "residuals": m, "operands": ops, "ms_per_pass": ms_pass, "active_W": round(W, 1), "all_samples_mean_W": round(sum(pw) / len(pw), 1), "min_W": min(pw), "max_W": max(pw), "samples": len(pw),
                 "samples_W": pw, "sclk_MHz": round(sum(ck) / max(len(ck), 1)), "J_per_pass": round(W * ms_pass * 1e-3, 2),
                 "J_per_residual": round(W * ms_pass * 1e-3 / m, 3), "mfma_32x32x64_equivalents": mfma, "nJ_per_mfma": round(W * ms_pass * 1e-3 / mfma * 1e9, 2)})
    print(json.dumps(rows[-1]), flush=True)
# the same instruction with its operands in registers
exe = os.path.join(ROOT, "build", "mfma_rate")
os.makedirs(os.path.dirname(exe), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "mfma_rate.hip"), "-o", exe], check=True)
for shape, name in ((5, "FP4xFP6 16x16x128, 2 waves/SIMD"), (1, "FP4xFP6 32x32x64, 2 waves/SIMD"), (3, "FP4xFP4 32x32x64, 2 waves/SIMD")):
    pw, ck, txt = watch([exe, "20000", "1", str(shape), "7"], settle=1.5)
    mt = re.search(r"([0-9.]+) ms\s+([0-9.e+]+) MFMA/s\s+in-kernel clock (\d+) MHz", txt)
    if not pw or not mt:
        rows.append({"workload": "mfma_rate " + name, "error": txt[-300:]}); continue
    W = active_mean(pw)
    rate = float(mt.group(2))
    rows.append({"workload": "tools/mfma_rate (operands in registers, dosage-like x digits): " + name, "active_W": round(W, 1), "samples": len(pw), "samples_W": pw,
                 "mfma_per_s": rate, "in_kernel_clock_MHz": int(mt.group(3)), "sclk_MHz": round(sum(ck) / max(len(ck), 1)), "nJ_per_mfma": round(W / rate * 1e9, 2)})
    print(json.dumps(rows[-1]), flush=True)
json.dump({"idle_W": idle, "rows": rows, "how": "rocm-smi --showpower / --showclocks sampled beside a looping child (first 2 s dropped)"}, open(out_file, "w"), indent=1)
