"""Per-step view of a rocprofv3 --kernel-trace of bench.py: the kernels between two X'r passes, in order, with durations and gaps.
usage: python tools/trace_chain.py <dir with *_kernel_trace.csv> [steps from the end]"""
import csv, glob, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_xtv_dma" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1_000_000]
a, b = idx[-back - 1], idx[-1]
prev_end, tot, first = None, 0.0, None
for r in rows[a:b + 1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mih::", "")[:44]
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    big = en - st > 1_000_000
    if big and first is not None:
        print(f"   -> between the passes: {(st - first) / 1e3:8.1f} us\n")
    print(f"{name:46s} {(en - st) / 1e3:10.1f} us   gap {gap:6.1f} us")
    if big:
        first = en
    prev_end = en
