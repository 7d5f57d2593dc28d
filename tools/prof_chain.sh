#!/bin/bash
# kernel trace of the driver's bench command (20 steps) and the per-step chain between two X'r passes; then the bench itself, unprofiled.
# usage: tools/prof_chain.sh TAG [pytest -k expression]
tag=$1; sel=${2:-"resident or g1_golden or session_run"}
R=$GRAFT_REPO_ROOT
cd $R && timeout 900 python -m pytest tests -x -q -m gpu -k "$sel" 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${tag}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_stats -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cv --no-mv > $R/gpurun_out/prof_${tag}_stats.json 2> $R/gpurun_out/prof_${tag}_stats.err
python3 $R/tools/trace_chain.py $R/gpurun_out/prof_${tag}_stats 2
cd $R && python bench.py --no-cv --no-mv --no-cpu-baseline --steps 100 > gpurun_out/${tag}_bench100.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench100.json").read().strip().splitlines()[-1])
print("bench 100 steps:", d["value"], "it/s", d["ms_per_step"], "ms/step; outside the pass", d["config"].get("host_small_kernels_and_exchange_ms_per_step"), "ms; backtracks", d["config"]["backtracks_in_timed_steps"], "kernel_ms", d["roofline"]["kernel_ms"], "launches", d["roofline"]["launches"])
PY
