#!/bin/bash
# Counter passes over the bench's resident step chain (20 steps, no CV / multivariate legs): per kernel of the chain, counters per LIVE launch
# (launches whose gate was closed are told apart by their wave cycles).  usage: tools/pmc_chain.sh [kernel name pattern, default k_res_]
pat=${1:-k_res_}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_chain_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace -d $out -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cv --no-mv > $R/gpurun_out/pmc_chain_$i.log 2>&1
  echo "set $i: rc=$?"
done
python3 - <<PY
import sqlite3, glob, collections
for i in (1, 2, 3):
    dbs = glob.glob("$R/gpurun_out/pmc_chain_%d/*.db" % i)
    if not dbs: continue
    c = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event_")][0]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc_")][0]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
    q = f"select s.kernel_name, i.name, d.id, sum(e.value), d.end - d.start from {pmc} e join {info} i on e.pmc_id = i.id join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id where s.kernel_name like '%$pat%' or s.kernel_name like '%k_digits%' or s.kernel_name like '%k_xtv_finalize%' group by d.id, i.name"
    per = collections.defaultdict(list)
    for name, cname, did, val, dur in c.execute(q):
        per[(name.split("(")[0].replace("void ", "").replace("mih::", "")[:28], cname)].append((val, dur))
    for (name, cname), rows in sorted(per.items()):
        live = [r for r in rows if r[1] > 6000]          # (a launch whose gate was closed lasts ~4-5 us under the profiler)
        if not live: continue
        print(f"set {i} {name:28s} {cname:22s} {sum(r[0] for r in live) / len(live):14.1f}   live launches {len(live):3d}  mean {sum(r[1] for r in live) / len(live) / 1e3:6.1f} us")
PY
