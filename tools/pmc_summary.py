"""Summarise rocprofv3 --pmc results (rocpd sqlite, ROCm 7.2 default output): per kernel matching a substring, the mean
counter value (summed over the counter's instances) and the mean duration per launch.
usage: pmc_summary.py KERNEL_SUBSTRING results.db [results2.db ...]  -> JSON"""
import json, sqlite3, sys
sub = sys.argv[1]
res = {}
for db in sys.argv[2:]:
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, name, counter_name, counter_value, duration from pmc_events").fetchall()
    per = {}
    dur = {}
    for did, kname, cname, val, d in rows:
        if sub not in kname:
            continue
        per.setdefault(cname, {}).setdefault(did, 0.0)
        per[cname][did] += float(val)
        dur[did] = float(d)
    for cname, v in per.items():
        vals = list(v.values())
        res[cname] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals),
                      "mean_kernel_ms": sum(dur[k] for k in v) / len(v) / 1e6}
print(json.dumps(res, indent=1))
