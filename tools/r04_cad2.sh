#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04; mkdir -p $O; cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off > $O/cad2_a.json 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off > $O/cad2_b.json 2>/dev/null
python3 bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off > $O/cad2_c.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof_cad2 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off > $O/cad2_trace.json 2>/dev/null
echo ok
