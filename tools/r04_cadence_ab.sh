#!/bin/bash
# GPU box: the headline at the driver's cadence with and without an option set, alternating (bash tools/r04_cadence_ab.sh "6=4" [reps])
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
args=""; for kv in $1; do args="$args --opt $kv"; done
for rep in $(seq 1 ${2:-2}); do
  for cfg in base new; do
    a=""; [ $cfg = new ] && a="$args"
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off $a 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
  done
done
