#!/bin/bash
# tools/ab_bench_libs.sh OLD.so -- the headline bench line (fit only) with the tree's library, with OLD.so in its place, and with the tree's
# again: same box, separate processes.  Box-to-box spread is 2-3 %: a change of the library is only visible in such an A/B/A.
cd $GRAFT_REPO_ROOT
old=$1
cp mendeliht.jl_amd/libmendeliht_hip.so /tmp/new.so
one() { python bench.py --no-cv --no-mv --no-dense --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],2), 'it/s', round(d['ms_per_step'],3), 'ms/step, kernel', round(d['roofline']['kernel_ms'],3), 'ms =', round(d['roofline']['frac'],4))"; }
one new
cp $old mendeliht.jl_amd/libmendeliht_hip.so; one old
cp /tmp/new.so mendeliht.jl_amd/libmendeliht_hip.so; one new
cp $old mendeliht.jl_amd/libmendeliht_hip.so; one old
cp /tmp/new.so mendeliht.jl_amd/libmendeliht_hip.so
