#!/bin/bash
# Package power and shader clock (rocm-smi, ~3 samples/s) while one kernel shape of the fused 12-residual pass loops for a
# few seconds.  usage: tools/power_sweep.sh OUT VARIANT[:RESIDUALS] ...      (e.g. 20 30 31 0:3)
out=$1; shift
R=${GRAFT_REPO_ROOT:-.}
: > $out
for spec in "$@"; do
  mv=${spec%%:*}; mm=12; [[ $spec == *:* ]] && mm=${spec##*:}
  python3 $R/tools/spin_multi.py $mm $mv 6 > /tmp/spin_$mv.log 2>&1 &
  pid=$!
  # wait for the matrix
  for i in $(seq 1 100); do grep -q ready /tmp/spin_$mv.log 2>/dev/null && break; sleep 0.2; done
  sleep 1.5
  pw=(); ck=()
  while kill -0 $pid 2>/dev/null; do
    line=$(rocm-smi --showpower --showclocks 2>/dev/null | tr '\n' ' ')
    p=$(echo "$line" | grep -o 'Power (W): [0-9.]*' | head -1 | awk '{print $3}')
    c=$(echo "$line" | grep -o 'sclk clock level: [0-9]*: ([0-9]*Mhz)' | head -1 | grep -o '[0-9]*Mhz')
    [[ -n $p ]] && echo "variant $mv m=$mm power_W $p sclk $c" >> $out
  done
  wait $pid
  echo "variant $mv m=$mm passes: $(grep ms/pass /tmp/spin_$mv.log | tail -3 | tr '\n' ' ')" >> $out
done
