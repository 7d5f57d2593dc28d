cd $GRAFT_REPO_ROOT
MENDELIHT_CV_TRACE=1 timeout 300 python tools/bench_cv100.py 2> gpurun_out/r05_cv_trace.txt | tail -4
grep "iterations" gpurun_out/r05_cv_trace.txt | sort -u | awk '{print $4, $6, $7}' | sort -n | awk '{printf "%s:k%s:%s ", $1,$2,$3} END {print ""}' | head -c 3000
grep -c "round" gpurun_out/r05_cv_trace.txt
