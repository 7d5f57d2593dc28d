cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_cv -o cv -- python3 $R/tools/bench_cv100.py > $R/gpurun_out/prof_r02_cv.log 2>&1
grep "k_xtv_dma16" $R/gpurun_out/prof_r02_cv/cv_kernel_stats.csv | awk -F'",' '{print $1, $2, $3, $4}' | cut -c1-60,140-
