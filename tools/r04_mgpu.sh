#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest6.log 2>&1; tail -3 gpurun_out/r04/gputest6.log
python3 bench.py --rccl-rehearsal --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 > gpurun_out/r04/bench_rccl1.json 2> gpurun_out/r04/bench_rccl1.err || echo rccl-fail
python3 -c "
import json
d=json.load(open('gpurun_out/r04/bench_rccl1.json'))
a=d['assembly']; print('carve ms',d['ms_per_step'],'with assembly',a['ms_per_step'],'int8 wire',a['int8_wire']['ms_per_step'],'gather_to_host',a.get('gather_to_host_ms'))
s=d['strong']; print('strong',s['ms_per_step'],s['ms_per_step_with_assembly'])
"
