#!/bin/bash
# The driver's bench command under rocprofv3: kernel-trace statistics, then FETCH_SIZE and WRITE_SIZE in separate passes.
# usage: tools/prof_bench.sh TAG   -> gpurun_out/prof_TAG_{stats,fetch,write}/
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_${tag}_stats $R/gpurun_out/prof_${tag}_fetch $R/gpurun_out/prof_${tag}_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_stats -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cv > $R/gpurun_out/prof_${tag}_stats.json 2> $R/gpurun_out/prof_${tag}_stats.err
echo "stats rc=$?"; cat $R/gpurun_out/prof_${tag}_stats.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_${tag}_fetch -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cv > $R/gpurun_out/prof_${tag}_fetch.json 2> $R/gpurun_out/prof_${tag}_fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_${tag}_write -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cv > $R/gpurun_out/prof_${tag}_write.json 2> $R/gpurun_out/prof_${tag}_write.err
echo "write rc=$?"
find $R/gpurun_out/prof_${tag}_stats -name "*.csv" | head
python3 $R/tools/traffic_from_rocpd.py "k_xtv_dma<1, 2, 4, 8, false" "k_xtv_dma<1,2,4,8,fp4>" $(find $R/gpurun_out/prof_${tag}_fetch -name "*results.db" | head -1) $(find $R/gpurun_out/prof_${tag}_write -name "*results.db" | head -1) 500000 1000000 > $R/gpurun_out/${tag}_traffic.json
cat $R/gpurun_out/${tag}_traffic.json
