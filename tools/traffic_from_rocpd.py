"""profiles/rNN_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite output) of the bench
command (tools/prof_bench.sh).  Counters are in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced streaming read on gfx950, so it is doubled.
usage: traffic_from_rocpd.py PROFILER_KERNEL_SUBSTRING LIBRARY_KERNEL_NAME fetch/pmc_results.db write/pmc_results.db N P"""
import json, sqlite3, sys

sub, libname, fdb, wdb, n, p = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])


def mean_counter(db, counter):
    cur = sqlite3.connect(db).cursor()
    per, names = {}, set()
    for did, kname, cname, val, dur in cur.execute("select dispatch_id, name, counter_name, counter_value, duration from pmc_events"):
        # (round 5) launches of a device-resident fit are gated: one queued behind a step that turned out to need backtracking
        # does nothing (a few microseconds, no traffic) -- only launches that ran (> 1 ms) are averaged
        if sub in kname and cname == counter and float(dur) > 1e6:
            per[did] = per.get(did, 0.0) + float(val)
            names.add(kname.split("(")[0])
    return sum(per.values()) / len(per), len(per), sorted(names)


f, nf, names = mean_counter(fdb, "FETCH_SIZE")
w, nw, _ = mean_counter(wdb, "WRITE_SIZE")
hbm = 2.0 * f * 1024 + w * 1024
alg = p * ((n + 3) // 4) + 8.0 * (n + p) + 16.0 * p
print(json.dumps({
    "_how": "rocprofv3 --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE around `python3 bench.py --steps 3 --warmup 1 "
            "--no-cpu-baseline --no-cv` (tools/prof_bench.sh; rocpd sqlite output); counters are in KiB; per MI355X_MICROARCH.md "
            "(HBM section) FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read on gfx950, so it is doubled.",
    "workload": {"n": n, "p": p},
    "kernel": libname,
    "profiler_kernel_name": names,
    "launches_averaged": [nf, nw],
    "FETCH_SIZE_KiB_avg_per_launch": f,
    "WRITE_SIZE_KiB_avg_per_launch": w,
    "hbm_bytes_per_launch": hbm,
    "algorithmic_bytes_per_launch": alg,
    "ratio_to_algorithmic": hbm / alg}, indent=1))
