"""Derived figures from the counter passes of tools/pmc_fused.sh (rocpd sqlite files): per launch of the kernel whose
name contains SUBSTRING -- counters summed over their instances, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration),
matrix-pipe busy share, LDS busy share, wave-cycle split, HBM traffic (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes: the
gfx950 correction of MI355X_MICROARCH.md) and L2 hit rate.
usage: pmc_report.py SUBSTRING gpurun_out/pmc_TAG_*/pmc_results.db > profiles/r02_pmc_TAG.json"""
import json, sqlite3, sys

sub, dbs = sys.argv[1], sys.argv[2:]
c, dur, names = {}, {}, set()
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    per, d = {}, {}
    for did, kname, cname, val, dd in cur.execute("select dispatch_id, name, counter_name, counter_value, duration from pmc_events"):
        if sub in kname:
            per.setdefault(cname, {}).setdefault(did, 0.0)
            per[cname][did] += float(val)
            d[did] = float(dd) / 1e6
            names.add(kname.split("(")[0])
    for cname, v in per.items():
        c[cname] = sum(v.values()) / len(v)
        dur[cname] = sum(d[k] for k in v) / len(v)
SIMDS, CUS, XCDS = 1024, 256, 8
out = {"kernel": sorted(names), "counters_mean_per_launch": c, "kernel_ms_in_that_pass": dur, "derived": {}}
g = out["derived"]
if "GRBM_GUI_ACTIVE" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / XCDS
    g["kernel_cycles"] = cyc
    g["effective_clock_GHz"] = cyc / (dur["GRBM_GUI_ACTIVE"] * 1e-3) / 1e9
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        g["matrix_pipe_busy_share"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc)
        if "SQ_INSTS_MFMA" in c:
            g["busy_cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"]
    if "SQ_LDS_IDX_ACTIVE" in c:
        cyc2 = cyc * dur["SQ_LDS_IDX_ACTIVE"] / dur["GRBM_GUI_ACTIVE"]
        g["lds_busy_share"] = c["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc2)
if "SQ_WAVE_CYCLES" in c:
    wc = c["SQ_WAVE_CYCLES"]
    for k, nm in (("SQ_ACTIVE_INST_ANY", "wave_cycles_issuing"), ("SQ_WAIT_INST_ANY", "wave_cycles_issue_stalled"), ("SQ_WAIT_ANY", "wave_cycles_parked_at_waitcnt_or_barrier")):
        if k in c:
            g[nm] = c[k] / wc
if "SQ_INSTS_MFMA" in c:
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD"):
        if k in c:
            g[k.lower() + "_per_mfma"] = c[k] / c["SQ_INSTS_MFMA"]
if "SQ_INSTS_LDS" in c and "SQ_INSTS_MFMA" in c:
    g["sq_insts_lds_per_mfma"] = c["SQ_INSTS_LDS"] / c["SQ_INSTS_MFMA"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    g["hbm_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
    g["achieved_GBps_of_that_traffic"] = g["hbm_bytes_per_launch"] / (dur["FETCH_SIZE"] * 1e-3) / 1e9
if "TCC_HIT_sum" in c:
    g["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
print(json.dumps(out, indent=1))
