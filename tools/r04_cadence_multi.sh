#!/bin/bash
# GPU box: the headline at the driver's cadence for several option sets, interleaved (bash tools/r04_cadence_multi.sh reps "set1" "set2" ...; "-" = defaults)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
reps=$1; shift
for rep in $(seq 1 $reps); do
  for spec in "$@"; do
    a=""; if [ "$spec" != "-" ]; then for kv in $spec; do a="$a --opt $kv"; done; fi
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off $a 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$spec'.ljust(16), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), flush=True)"
  done
done
