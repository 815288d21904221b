#!/bin/bash
# GPU box: the headline at the driver's cadence for several builds of the library, interleaved (bash tools/r04_cadence_libs.sh reps lib1 lib2 ...)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
reps=$1; shift
for rep in $(seq 1 $reps); do
  for lib in "$@"; do
    SPACECARVE_LIB=$R/$lib python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib'.ljust(22), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), flush=True)"
  done
done
