#!/bin/bash
# Round 5's profile set in one GPU call (tools/README.md):
#  1. the driver's bench command WITH the cv_iht and multivariate legs under rocprofv3 --kernel-trace --stats
#     -> gpurun_out/r05_bench_kernel_stats.csv (+ its own JSON line), and the per-step chain of the resident fit (tools/trace_chain.py)
#  2. FETCH_SIZE / WRITE_SIZE passes of the bench command -> gpurun_out/r05_traffic.json
#  3. counter passes (stall / issue split, MFMA, LDS, clock) over the SINGLE-FIT pass k_xtv_dma<1,2,4,8,fp4> -> gpurun_out/r05_pmc_dma_single.json
# The profiled program is `python3 <script>` directly behind `--` (no env / shell hop).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_r05_stats $R/gpurun_out/prof_r05_fetch $R/gpurun_out/prof_r05_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05_stats -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r05_bench_under_rocprof.json 2> $R/gpurun_out/prof_r05_stats.err
echo "stats rc=$?"
cp $(find $R/gpurun_out/prof_r05_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05_bench_kernel_stats.csv
python3 $R/tools/trace_chain.py $R/gpurun_out/prof_r05_stats 2 any > $R/gpurun_out/r05_step_chain_with_legs.txt 2>&1; tail -5 $R/gpurun_out/r05_step_chain_with_legs.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_r05_fetch -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cv --no-mv > $R/gpurun_out/prof_r05_fetch.json 2> $R/gpurun_out/prof_r05_fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_r05_write -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cv --no-mv > $R/gpurun_out/prof_r05_write.json 2> $R/gpurun_out/prof_r05_write.err
echo "write rc=$?"
python3 $R/tools/traffic_from_rocpd.py "k_xtv_dma<1, 2, 4, 8, false" "k_xtv_dma<1,2,4,8,fp4>" $(find $R/gpurun_out/prof_r05_fetch -name "*results.db" | head -1) $(find $R/gpurun_out/prof_r05_write -name "*results.db" | head -1) 500000 1000000 > $R/gpurun_out/r05_traffic.json
cat $R/gpurun_out/r05_traffic.json | head -20
cd $R && bash tools/pmc_fused.sh r05single 0 1
python3 tools/pmc_report.py "k_xtv_dma<1" $(ls -d gpurun_out/pmc_r05single_*/ | sed 's|/$||' | while read d; do find $d -name "*results.db" | head -1; done) > gpurun_out/r05_pmc_dma_single.json
cat gpurun_out/r05_pmc_dma_single.json | head -60
