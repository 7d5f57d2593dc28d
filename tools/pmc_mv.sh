#!/bin/bash
# Counter passes over one multivariate fit at configs[4] size (tools/prof_mv.py): the X*B kernel k_xv_snp_cached_mt<10>.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_mv_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace -d $out -o pmc -- python3 $R/tools/prof_mv.py > $R/gpurun_out/pmc_mv_$i.log 2>&1
  echo "set $i: rc=$?"
done
python3 - <<PY
import sqlite3, glob, collections
for i in (1, 2, 3):
    dbs = glob.glob("$R/gpurun_out/pmc_mv_%d/*.db" % i)
    if not dbs: continue
    c = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event_")][0]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc_")][0]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
    q = f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id), sum(d.end - d.start) from {pmc} e join {info} i on e.pmc_id = i.id join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id where s.kernel_name like '%k_xv_snp_cached_mt%' group by i.name"
    try:
        for row in c.execute(q):
            print(i, row[0][:40], row[1], row[2] / max(row[3], 1), "launches", row[3])
    except Exception as e:
        print("query failed", e, [t for t in tabs if "pmc" in t][:5])
PY
