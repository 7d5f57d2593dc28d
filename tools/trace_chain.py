"""Per-step view of a rocprofv3 --kernel-trace of bench.py: the kernels between two X'r passes, in order, with durations and gaps.
usage: python tools/trace_chain.py <dir with *_kernel_trace.csv> [steps to show] [resident|host|any]
`resident` (default) shows steps of the device-resident chain (those with a k_res_decide in them), `host` the host-driven ones
(bench.py's same-box A/B runs them after the timed region), `any` the last ones of the trace whatever they are."""
import csv, glob, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kind = sys.argv[3] if len(sys.argv) > 3 else "resident"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
idx = [i for i, r in enumerate(rows) if "k_xtv_dma" in r["Kernel_Name"] and dur(r) > 1_000_000]
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    names = [r["Kernel_Name"] for r in rows[a + 1:b]]
    res = any("k_res_decide" in x for x in names)
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 5_000_000: continue      # (set-up between two fits, not a step)
    if kind == "any" or (kind == "resident") == res: steps.append((a, b))
for a, b in steps[-back:]:
    prev_end, first = None, None
    empty = 0
    for r in rows[a:b + 1]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mih::", "")[:44]
        gap = (st - prev_end) / 1e3 if prev_end else 0.0
        big = en - st > 1_000_000
        if big and first is not None:
            print(f"   -> between the passes: {(st - first) / 1e3:8.1f} us ({b - a - 1} launches)\n")
        print(f"{name:46s} {(en - st) / 1e3:10.1f} us   gap {gap:6.1f} us")
        if big: first = en
        prev_end = en
# mean over all steps of the kind
if steps:
    tot = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 for a, b in steps]
    print(f"{len(steps)} {kind} steps in the trace: mean {sum(tot) / len(tot):.1f} us between the passes (min {min(tot):.1f}, max {max(tot):.1f}); "
          f"mean launches {sum(b - a - 1 for a, b in steps) / len(steps):.1f}")
