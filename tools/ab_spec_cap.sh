#!/bin/bash
# (round 5, measurement build) how many attempt slots beyond the first a series of the resident chain may hold: chain time and short series per cap
cd $GRAFT_REPO_ROOT
for cap in 2 1 0 3; do for rep in 1 2; do
  MENDELIHT_HIP_PROBES=1 MENDELIHT_SPEC_CAP=$cap python bench.py --no-cv --no-mv --no-cpu-baseline --steps 100 --warmup 5 2>/dev/null > /tmp/cap.json
  python - "$cap" <<PY
import json, sys
d = json.loads(open("/tmp/cap.json").read().strip().splitlines()[-1]); c = d["config"]
print("cap", sys.argv[1], "chain_ms_per_step", round(c["chain_ms_per_step"], 4), "short series", c["resident_steps"]["resident_attempts"], "kernel_ms", round(d["roofline"]["kernel_ms"], 3))
PY
done; done
