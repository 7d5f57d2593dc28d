"""Print the kernels (and the idle gaps) between two consecutive X'r passes from a rocprofv3 kernel trace.
usage: timeline.py <kernel_trace.csv> [which interval, default: the shortest]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ix = [i for i, r in enumerate(rows) if "k_xtv_mfma" in r["Kernel_Name"] or "k_xtv_dma" in r["Kernel_Name"]]
spans = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]), a, b) for a, b in zip(ix[:-1], ix[1:])]
spans.sort()
span, a, b = spans[int(sys.argv[2])] if len(sys.argv) > 2 else spans[0]
prev = int(rows[a]["End_Timestamp"])
ksum = gsum = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"gap {(s - prev) / 1e3:7.1f} us  run {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'].split('(')[0][-44:]}")
    if r is not rows[b]:
        ksum += e - s
    gsum += s - prev
    prev = e
print(f"span {span / 1e3:.1f} us = kernels {ksum / 1e3:.1f} us + gaps {gsum / 1e3:.1f} us; all spans (us): {[round(x[0] / 1e3) for x in spans]}")
