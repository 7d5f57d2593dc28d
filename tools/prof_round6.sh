#!/bin/bash
# tools/prof_round6.sh -- the round's profile set in one gpurun call: the bench under rocprofv3 (kernel stats, traffic counters),
# the counter passes of the fused 19-residual pass, the energy table, the 10^4 repetition stress.
cd $GRAFT_REPO_ROOT
bash tools/prof_bench.sh r06 2>&1 | tail -15
bash tools/pmc_fused.sh r06m19 0 19 2>&1 | tail -8
python3 tools/pmc_report.py "k_xtv_dma16<6" $(find gpurun_out/pmc_r06m19_* -name "*results.db") > gpurun_out/r06_pmc_dma16_m19.json 2> gpurun_out/r06_pmc_summary.err; tail -3 gpurun_out/r06_pmc_summary.err
python3 tools/energy_table.py gpurun_out/r06_energy.json 2>&1 | tail -14
python3 tools/stress_chain.py 10000 > gpurun_out/r06_stress_chain.json 2> gpurun_out/r06_stress_chain.err; cat gpurun_out/r06_stress_chain.json | cut -c1-600
