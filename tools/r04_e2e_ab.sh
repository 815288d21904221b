#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python3 tools/r04_e2e_ab.py 0 1 12 2>/dev/null | tail -1
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --skip-other-path --traffic-passes off --cold-reps 0 --e2e-reps 7 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench e2e', d['e2e']['ms_all'])"
python3 tools/r04_e2e_ab.py 0 1 12 2>/dev/null | tail -1
python3 tools/r04_e2e_ab.py 0 0 12 2>/dev/null | tail -1
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
