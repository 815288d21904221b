#!/bin/bash
# GPU box: the literal scene now and then takes 1.4-1.9 ms per batch instead of 0.25 for a whole process -- catch one with its kernel breakdown
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for i in $(seq 1 ${1:-10}); do
  python3 tools/bench_scenes.py --steps 30 --scenes plant,dense,literal,noise --trace-steps 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
l=d['literal']
print('run $i literal', l['ms'], l['ms_host'], l.get('slowest_step'), flush=True)"
done
